# coding=utf-8
"""Pins the BACKWARD oracle to TensorFlow's documented gradient semantics.

The reference obtains its gradients from tf.GradientTape over nn/kernel/map_reduce.py / nn/kernel/segment.py:26-33
(training loops: demo/demo_gcn.py:68-77).  TensorFlow cannot run here, so:

  * oracle/tf_gradients.py restates TF's *registered* gradient functions (math_grad.py: _UnsortedSegmentSumGrad,
    _UnsortedSegmentMinOrMaxGrad, _GatherV2Grad, ...) and chains them along the reference's forward lines;
  * THIS file holds that restatement to HAND-DERIVED known answers — tied maxima share the gradient evenly, an empty
    segment contributes nothing, the softmax gradient under stop_gradient(max) is the plain softmax Jacobian
    p_i (delta_ij - p_j), the +1e-8 in the denominator leaves a 1e-8 derivative on a single-edge segment, duplicate
    gather indices sum, negative segment ids are dropped;
  * and checks that the float64 torch-autograd restatement the -m gpu tests compare the HIP kernels with
    (tests/test_gpu_backward.py::_ref_aggregate) computes the same gradients on random graphs with ties and empty rows.
tests/test_gpu_backward.py::test_hip_gradients_match_tf_registered_gradients then compares the HIP backward with
oracle/tf_gradients.py directly.  No GPU here."""
import numpy as np
import pytest
import torch

from oracle import tf_gradients as G


def test_kat_tied_maxima_split_evenly():
    """node 0 <- {1, 2, 3} with x = 5, 5, 1: two tied maxima take 1/2 each; node 1 <- {3}: the only entry takes all."""
    ei = np.array([[0, 0, 0, 1], [1, 2, 3, 3]], np.int32)
    x = np.array([[0.0], [5.0], [5.0], [1.0]])
    g = np.array([[1.0], [1.0], [0.0], [0.0]])
    out, dx, _ = G.aggregate_neighbors_grad(x, ei, None, "max", "identity", g)
    assert out[0, 0] == 5.0 and out[1, 0] == 1.0
    assert out[2, 0] == G.FLT_LOWEST and out[3, 0] == G.FLT_LOWEST            # empty segments: float32 lowest
    assert np.array_equal(dx[:, 0], [0.0, 0.5, 0.5, 1.0])
    # three-way tie with a weighted mapper: msg = w * x = 6 for all three edges -> 1/3 each, times w into x, times x into w
    ei = np.array([[0, 0, 0], [1, 2, 3]], np.int32)
    x = np.array([[9.0], [3.0], [2.0], [6.0]])
    w = np.array([2.0, 3.0, 1.0])
    out, dx, dw = G.aggregate_neighbors_grad(x, ei, w, "max", "identity", np.array([[3.0], [0], [0], [0]]))
    assert out[0, 0] == 6.0
    assert np.allclose(dx[:, 0], [0.0, 2.0, 3.0, 1.0]) and np.allclose(dw, [3.0, 2.0, 6.0])
    # duplicated edge (the same source twice): both copies are "selected", each gets 1/2, the source collects both halves
    ei = np.array([[0, 0], [1, 1]], np.int32)
    _, dx, _ = G.aggregate_neighbors_grad(np.array([[0.0], [4.0]]), ei, None, "max", "identity", np.array([[1.0], [0.0]]))
    assert np.array_equal(dx[:, 0], [0.0, 1.0])


def test_kat_empty_segment_and_dropped_ids():
    # unsorted_segment_max gradient: segment 1 is empty (0/0 never gathered -> no NaN), id -1 is dropped (gets 0)
    data = np.array([[1.0, 7.0], [3.0, 2.0], [9.0, 9.0]])
    ids = np.array([0, 0, -1])
    out = np.array([[3.0, 7.0], [G.FLT_LOWEST, G.FLT_LOWEST]])
    d = G.unsorted_segment_max_grad(np.array([[10.0, 20.0], [30.0, 40.0]]), data, ids, out)
    assert np.array_equal(d, [[0.0, 20.0], [10.0, 0.0], [0.0, 0.0]]) and np.isfinite(d).all()
    assert np.array_equal(G.unsorted_segment_sum_grad(np.array([[1.0], [2.0]]), np.array([1, -1, 0, 1])),
                          [[2.0], [0.0], [1.0], [2.0]])
    # mean: divisor max(count, 1); segment 2 of 3 is empty
    d = G.unsorted_segment_mean_grad(np.array([[6.0], [5.0], [4.0]]), np.array([0, 0, 0, 1]), 3)
    assert np.allclose(d[:, 0], [2.0, 2.0, 2.0, 5.0])
    # gather: duplicate indices sum (IndexedSlices densified)
    assert np.array_equal(G.gather_grad(np.array([[1.0], [2.0], [4.0]]), np.array([2, 0, 2]), 4), [[2.0], [0.0], [5.0], [0.0]])


def test_kat_mean_and_sum_with_weights():
    """node 0 <- {1 (w=2), 2 (w=3)}; sum updater: out0 = x0 + (2 x1 + 3 x2) [sum] or x0 + (2 x1 + 3 x2)/2 [mean]."""
    ei = np.array([[0, 0], [1, 2]], np.int32)
    x = np.array([[1.0, 1.0], [2.0, -1.0], [4.0, 0.5]])
    w = np.array([2.0, 3.0])
    g = np.array([[1.0, 10.0], [0, 0], [0, 0]])
    out, dx, dw = G.aggregate_neighbors_grad(x, ei, w, "sum", "sum", g)
    assert np.allclose(out[0], [1 + 4 + 12, 1 - 2 + 1.5])
    assert np.allclose(dx, [[1, 10], [2, 20], [3, 30]]) and np.allclose(dw, [2 - 10, 4 + 5])
    out, dx, dw = G.aggregate_neighbors_grad(x, ei, w, "mean", "identity", g)
    assert np.allclose(out[0], [8.0, -0.25]) and np.allclose(dx, [[0, 0], [1, 10], [1.5, 15]]) and np.allclose(dw, [-4, 4.5])


def test_kat_softmax_under_stop_gradient_is_the_plain_softmax_jacobian():
    """One segment with scores (0, ln 3): p = (1/4, 3/4) (up to the 1e-8), dL/ds_j = p_j (g_j - sum_i p_i g_i)."""
    s = np.array([0.0, np.log(3.0)])
    score, ds = G.segment_softmax_grad(s, np.array([0, 0]), 1, np.array([1.0, 0.0]))
    assert np.allclose(score, [0.25, 0.75], atol=1e-8)
    assert np.allclose(ds, [0.1875, -0.1875], atol=1e-8)
    # general: any upstream gradient, two segments of different size and one empty segment
    rng = np.random.Generator(np.random.PCG64(0))
    s = rng.standard_normal(7) * 3
    ids = np.array([0, 2, 2, 0, 2, 0, 0])
    g = rng.standard_normal(7)
    score, ds = G.segment_softmax_grad(s, ids, 4, g)
    for seg in (0, 2):
        m = ids == seg
        p = np.exp(s[m] - s[m].max())
        p = p / (p.sum() + 1e-8)
        assert np.allclose(score[m], p, rtol=1e-12)
        assert np.allclose(ds[m], p * (g[m] - (p * g[m]).sum()), rtol=1e-9, atol=1e-15)
    # 2-D data (one column per head), as gat.py:83-84 uses it through the [H*E] flattening
    s2 = rng.standard_normal((7, 3))
    g2 = rng.standard_normal((7, 3))
    sc2, ds2 = G.segment_softmax_grad(s2, ids, 4, g2)
    for h in range(3):
        sc1, ds1 = G.segment_softmax_grad(s2[:, h], ids, 4, g2[:, h])
        assert np.allclose(sc2[:, h], sc1) and np.allclose(ds2[:, h], ds1)


def test_kat_the_1e8_term():
    """A single-edge segment: e = 1, denominator 1 + 1e-8, p = 1/(1+1e-8), dp/ds = p (1 - p) = 1e-8/(1+1e-8)^2 — not 0
    in exact arithmetic (and exactly 0 in fp32, where 1 + 1e-8 == 1: both are far inside the 1e-5 band)."""
    score, ds = G.segment_softmax_grad(np.array([2.5]), np.array([0]), 1, np.array([1.0]))
    assert score[0] == pytest.approx(1.0 / (1.0 + 1e-8), rel=1e-15)
    assert ds[0] == pytest.approx(1e-8 / (1.0 + 1e-8) ** 2, rel=1e-6)
    assert np.float32(1.0) + np.float32(1e-8) == np.float32(1.0)
    # stop_gradient changes the true derivative only through that term: compare with the derivative of the same formula
    # WITHOUT stop_gradient (finite differences of the full function) — difference O(1e-8)
    s = np.array([0.3, -1.2, 0.9])
    ids = np.zeros(3, np.int64)
    g = np.array([0.7, -0.4, 1.1])

    def f(v):
        e = np.exp(v - v.max())
        return float(((e / (e.sum() + 1e-8)) * g).sum())
    fd = np.array([(f(s + h) - f(s - h)) / 2e-6 for h in np.eye(3) * 1e-6])
    _, ds = G.segment_softmax_grad(s, ids, 1, g)
    assert np.allclose(ds, fd, atol=1e-7)


def test_kat_gcn_layer_path_graph():
    """3-node path 0-1-2, renormalised adjacency (weights 1/2, 1/sqrt6, 1/3 — the SURVEY.md 8c known answer), one
    feature, kernel [[2]], bias [-0.5], relu; upstream gradient 1 on every node."""
    s6 = 1 / np.sqrt(6.0)
    idx = np.array([[0, 0, 1, 1, 1, 2, 2], [0, 1, 0, 1, 2, 1, 2]], np.int32)
    w = np.array([0.5, s6, s6, 1 / 3.0, s6, s6, 0.5])
    x = np.array([[1.0], [-2.0], [0.5]])
    out, dx, dk, db = G.gcn_layer_grad(x, idx, w, np.array([[2.0]]), np.array([-0.5]), True, np.ones((3, 1)))
    h0 = 2 * x[:, 0]
    pre = np.array([0.5 * h0[0] + s6 * h0[1], s6 * h0[0] + h0[1] / 3 + s6 * h0[2], s6 * h0[1] + 0.5 * h0[2]]) - 0.5
    assert np.allclose(out[:, 0], np.maximum(pre, 0))
    live = (pre > 0).astype(np.float64)                       # only node 1... computed, not assumed
    a_t_g = np.array([0.5 * live[0] + s6 * live[1], s6 * live[0] + live[1] / 3 + s6 * live[2], s6 * live[1] + 0.5 * live[2]])
    assert np.allclose(dx[:, 0], 2 * a_t_g) and np.allclose(dk[0, 0], (x[:, 0] * a_t_g).sum()) and np.allclose(db[0], live.sum())


@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("weighted", [True, False])
def test_torch_autograd_restatement_equals_tf_registered_gradients(op, weighted):
    """The float64 torch-autograd restatement the GPU tests use (scatter_reduce 'amax' distributes evenly among ties,
    like TF) vs oracle/tf_gradients.py on graphs with duplicated edges (ties), quantised features (more ties) and
    empty rows."""
    from test_gpu_backward import _ref_aggregate
    rng = np.random.Generator(np.random.PCG64(5))
    n, e, f = 60, 500, 5
    ei = rng.integers(0, n - 6, size=(2, e)).astype(np.int32)
    ei = np.concatenate([ei, ei[:, :120]], axis=1)                               # duplicates
    x = np.round(rng.standard_normal((n, f)) * 2) / 2
    w = rng.integers(1, 3, ei.shape[1]).astype(np.float64) * 0.5 if weighted else None
    g = rng.standard_normal((n, f))
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = torch.tensor(w, dtype=torch.float64, requires_grad=True) if weighted else None
    ref = xr + _ref_aggregate(xr, ei, wr, op, n)
    ref.backward(torch.tensor(g))
    out, dx, dw = G.aggregate_neighbors_grad(x, ei, w, op, "sum", g)
    empty = np.bincount(ei[0], minlength=n) == 0
    assert np.allclose(out[~empty], ref.detach().numpy()[~empty], rtol=1e-12)
    assert np.allclose(dx, xr.grad.numpy(), rtol=1e-12, atol=1e-12)
    if weighted:
        assert np.allclose(dw, wr.grad.numpy(), rtol=1e-12, atol=1e-12)
