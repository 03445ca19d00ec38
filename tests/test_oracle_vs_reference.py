# coding=utf-8
"""Pins the oracle to the REFERENCE'S OWN PYTHON.

* ``test_oracle_matches_reference_golden``: oracle/tfg_oracle.py vs tests/golden/reference_cases.npz — the outputs the
  unmodified /root/reference/tf_geometric produced (tests/golden/make_golden_from_reference.py).  Runs anywhere.
* ``test_golden_file_is_what_the_reference_produces`` / ``test_oracle_equals_reference_python``: only where the
  reference checkout exists (this container): re-runs the reference live — the committed file must be reproduced
  bit for bit, and on further fuzz seeds the oracle must agree with the live reference.
No GPU, no product code: this file only relates the checker to the reference."""
import os
import types

import numpy as np
import pytest

from conftest import ROOT, assert_parity
import reference_cases as rc
from oracle.ref_harness import load_reference, reference_available

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_cases.npz")


def compare(case, got, golden, who):
    keys = [k for k in golden if k.startswith(case.name + "::")]
    assert keys, "no golden entries for " + case.name
    for full in keys:
        k = full[len(case.name) + 2:]
        if k not in got:
            if "::" in k:        # layer weights stored for the product executor
                continue
            raise AssertionError("{} did not produce {}::{}".format(who, case.name, k))
        a, b = np.asarray(got[k]), golden[full]
        assert a.shape == b.shape, "{}::{} shape {} vs reference {}".format(case.name, k, a.shape, b.shape)
        if k in case.exact:
            assert np.array_equal(a, b), "{}::{} must be bit-identical to the reference".format(case.name, k)
        else:
            assert_parity(a, b, tol=case.tol_of(k), what="{} {}::{}".format(who, case.name, k))


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLDEN))


@pytest.mark.parametrize("case", [c for c in rc.CASES if c.orc is not None], ids=lambda c: c.name)
def test_oracle_matches_reference_golden(case, oracle, golden):
    compare(case, case.orc(oracle, case.inputs()), golden, "oracle")


needs_ref = pytest.mark.skipif(not reference_available(), reason="/root/reference is not present on this machine")


@pytest.fixture(scope="module")
def R():
    tfg, tf, tfs, backend = load_reference()
    return types.SimpleNamespace(tfg=tfg, tf=tf, tfs=tfs, backend=backend)


@needs_ref
def test_reference_is_loaded_unmodified(R):
    import hashlib
    assert os.path.realpath(R.tfg.__file__).startswith(os.path.realpath("/root/reference"))
    for rel in ("nn/kernel/map_reduce.py", "nn/kernel/segment.py", "nn/conv/gcn.py", "nn/conv/gat.py",
                "nn/conv/graph_sage.py", "utils/graph_utils.py", "layers/conv/gcn.py", "layers/conv/gat.py",
                "layers/conv/graph_sage.py"):
        mod = "tf_geometric." + rel[:-3].replace("/", ".")
        import importlib
        m = importlib.import_module(mod)
        with open(os.path.join("/root/reference/tf_geometric", rel), "rb") as f:
            on_disk = hashlib.sha1(f.read()).hexdigest()
        with open(m.__file__, "rb") as f:
            assert hashlib.sha1(f.read()).hexdigest() == on_disk


@needs_ref
@pytest.mark.parametrize("case", rc.CASES, ids=lambda c: c.name)
def test_golden_file_is_what_the_reference_produces(case, R, golden):
    if str(golden["__backend__"]) != R.backend:
        pytest.skip("golden file was generated on backend {}".format(golden["__backend__"]))
    live = case.ref(R, case.inputs())
    for k, v in live.items():
        assert np.array_equal(np.asarray(v), golden["{}::{}".format(case.name, k)], equal_nan=True), \
            "{}::{}: committed golden differs from the live reference — regenerate".format(case.name, k)


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_reference_python(seed, R, oracle):
    """Fuzz: fresh graphs (duplicates, self-loops, isolated nodes), hot-path functions, oracle vs LIVE reference."""
    o, nn, tf = oracle, R.tfg.nn, R.tf
    g = rc.graph(60 + 17 * seed, 400 + 90 * seed, 5 + seed, seed=900 + seed, self_loops=seed, isolated=seed % 3)
    x, ei, w, n, rng = g["x"], g["ei"], g["w"], g["n"], g["rng"]
    gm = R.tfg.nn.conv.gcn.gcn_mapper
    for red in ("sum", "mean", "max"):
        ref = nn.aggregate_neighbors(x, ei, w, gm, getattr(nn, red + "_reducer"), nn.identity_updater).numpy()
        got = o.aggregate_neighbors(x, ei, w, o.gcn_mapper, getattr(o, red + "_reducer"), o.identity_updater)
        if red == "max":
            assert np.array_equal(got, ref)
        else:
            assert_parity(got, ref, what="aggregate " + red)
    f, u = x.shape[1], 4 + seed
    kernel, bias = rc.glorot(rng, f, u), rc.small_bias(rng, u)
    cfg = rc.NORM_CFGS[seed % len(rc.NORM_CFGS)]
    adj = R.tfs.SparseMatrix(ei, w, [n, n])
    assert_parity(o.gcn(x, ei, w, kernel, bias, "relu", **cfg),
                  nn.gcn(x, adj, kernel, bias, activation=tf.nn.relu, **cfg).numpy(), what="gcn {}".format(cfg))
    H = [1, 2, 4, 8, 2, 4][seed]
    A, U = H * (1 + seed % 3), H * 3
    wq, wk, wv = rc.glorot(rng, f, A), rc.glorot(rng, f, A), rc.glorot(rng, f, U)
    bq, bk, b = rc.small_bias(rng, A), rc.small_bias(rng, A), rc.small_bias(rng, U)
    assert_parity(o.gat(x, ei, wq, bq, "relu", wk, bk, "relu", wv, b, "relu", num_heads=H),
                  nn.gat(x, ei, wq, bq, tf.nn.relu, wk, bk, tf.nn.relu, wv, b, tf.nn.relu, num_heads=H).numpy(),
                  what="gat H={}".format(H))
    ws, wn, b2 = rc.glorot(rng, f, u), rc.glorot(rng, f, u), rc.small_bias(rng, 2 * u)
    for name in ("mean", "sum"):
        assert_parity(getattr(o, name + "_graph_sage")(x, ei, w, ws, wn, b2, "relu", normalize=bool(seed % 2)),
                      getattr(nn, name + "_graph_sage")(x, ei, w, ws, wn, b2, tf.nn.relu,
                                                        normalize=bool(seed % 2)).numpy(), what=name + " sage")
    s = (rng.standard_normal(ei.shape[1]) * 4).astype(np.float32)
    assert_parity(o.segment_softmax(s, ei[0], n), R.tfg.nn.kernel.segment.segment_softmax(s, ei[0], n).numpy(),
                  what="segment_softmax")


@needs_ref
def test_num_splits_rule_equals_reference(R):
    """dist.sharded.compute_num_or_size_splits (column-chunked halo) vs the reference's helper of the same name
    (utils/tf_sparse_utils.py:71-90): same split sizes, same refusals."""
    from tf_geometric_amd.dist.sharded import compute_num_or_size_splits as mine
    theirs = R.tfg.utils.tf_sparse_utils.compute_num_or_size_splits
    for f in (1, 7, 12, 100, 128, 602, 1433):
        for k in (None, 1, 2, 3, 4, 5, 7, 8, 16, 100):
            try:
                want = theirs(f, k)
            except Exception:
                want = "raises"
            try:
                got = mine(f, k)
            except Exception:
                got = "raises"
            if isinstance(want, list):
                want = [int(v) for v in want]
            assert got == want, (f, k, got, want)


def test_the_one_widened_tolerance_is_inherent_to_fp32(oracle, golden):
    """Every golden comparison uses the plain band 1e-5 + 1e-5*|ref| except propagation_convs::chebynet-None (2e-4).
    Evidence that this widening is a property of fp32, not of the HIP path: the REFERENCE'S OWN fp32 output (the golden
    vector) is itself several bands away from the float64-accumulated value of the same formula, while on every other
    output of the case it sits well inside one band."""
    case = rc.by_name("propagation_convs")
    exact = case.orc(oracle, case.inputs())
    widened = [k for k in exact if case.tol_of(k) != rc.TOL]
    assert widened == ["chebynet-None"]
    for k, v in exact.items():
        ref32 = golden["propagation_convs::" + k].astype(np.float64)
        v = np.asarray(v, np.float64)
        ratio = float((np.abs(ref32 - v) / (1e-5 + 1e-5 * np.abs(v))).max())
        if k in widened:
            assert ratio > 3.0, (k, ratio)          # measured 9.1: the reference itself leaves the plain band
        else:
            assert ratio < 0.5, (k, ratio)
    assert all(c.tol == rc.TOL for c in rc.CASES)
    assert sum(len(c.key_tol) for c in rc.CASES) == 1
