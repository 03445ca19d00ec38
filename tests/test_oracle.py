# coding=utf-8
"""Pins the CPU oracle (the reference ships no tests or golden vectors — SURVEY.md §4, §8c):
 (1) hand-derived known answers on the reference's own example graphs,
 (2) an independent second implementation (torch CPU index_add_/scatter_reduce, and the C restatement),
 (3) algebraic properties.  (The golden vectors — outputs of the reference's own Python — are in tests/golden/reference_cases.npz and
checked by tests/test_oracle_vs_reference.py.)"""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import assert_parity, ROOT


# ------------------------------------------------------------------ (1) known answers
def test_kat_tutorial_graph_sum_and_mean(oracle):
    """5-node weighted graph of tutorial_intro.py:24-30 (edge_index [[0,0,1,3],[1,2,2,1]], w [.9,.8,.1,.2])."""
    ei = np.array([[0, 0, 1, 3], [1, 2, 2, 1]], np.int32)
    w = np.array([0.9, 0.8, 0.1, 0.2], np.float32)
    x = np.arange(10, dtype=np.float32).reshape(5, 2)        # x[i] = [2i, 2i+1]
    s = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater)
    # row0 = .9*x1 + .8*x2 ; row1 = .1*x2 ; row3 = .2*x1 ; rows 2,4 empty -> 0
    exp = np.zeros((5, 2), np.float32)
    exp[0] = 0.9 * x[1] + 0.8 * x[2]
    exp[1] = 0.1 * x[2]
    exp[3] = 0.2 * x[1]
    assert_parity(s, exp, tol=1e-6, what="tutorial sum")
    m = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.mean_reducer, oracle.sum_updater)
    exp_m = x.copy()
    exp_m[0] += (0.9 * x[1] + 0.8 * x[2]) / 2
    exp_m[1] += 0.1 * x[2]
    exp_m[3] += 0.2 * x[1]
    assert_parity(m, exp_m, tol=1e-6, what="tutorial mean + sum_updater")
    mx = oracle.aggregate_neighbors(x, ei, None, oracle.identity_mapper, oracle.max_reducer, oracle.identity_updater)
    assert np.array_equal(mx[0], np.maximum(x[1], x[2])) and mx[2, 0] == oracle.FLT_LOWEST and mx[4, 1] == oracle.FLT_LOWEST


def test_kat_gcn_norm_path_graph(oracle):
    """Path 0-1-2 with self-loops: deg = (2,3,2) -> weights 1/sqrt6 on edges, 1/2, 1/3, 1/2 on the diagonal."""
    ei = np.array([[0, 1, 1, 2], [1, 0, 2, 1]], np.int32)
    idx, w = oracle.gcn_norm_adj(ei, None, 3)
    assert idx.tolist() == [[0, 1, 1, 2, 0, 1, 2], [1, 0, 2, 1, 0, 1, 2]]     # diagonal APPENDED (graph_utils.py:350-366)
    s = 1 / np.sqrt(6)
    assert_parity(w, np.array([s, s, s, s, 0.5, 1 / 3, 0.5], np.float32), tol=1e-7, what="renorm")
    _, w2 = oracle.gcn_norm_adj(ei, None, 3, renorm=False)                      # D^-1/2 A D^-1/2 + I, deg = (1,2,1)
    t = 1 / np.sqrt(2)
    assert_parity(w2, np.array([t, t, t, t, 1, 1, 1], np.float32), tol=1e-7, what="no renorm")
    _, w3 = oracle.gcn_norm_adj(ei, None, 3, improved=True)                     # fill 2 -> deg = (3,4,3)
    u = 1 / np.sqrt(12)
    assert_parity(w3, np.array([u, u, u, u, 2 / 3, 0.5, 2 / 3], np.float32), tol=1e-7, what="improved")
    _, w4 = oracle.gcn_norm_adj(ei, None, 3, norm="left")                       # rows of (A+I) sum to one
    assert_parity(w4, np.array([.5, 1 / 3, 1 / 3, .5, .5, 1 / 3, .5], np.float32), tol=1e-7, what="left")
    _, w5 = oracle.gcn_norm_adj(ei, None, 3, norm="right")                      # row degree of the COLUMN node
    assert_parity(w5, np.array([1 / 3, .5, .5, 1 / 3, .5, 1 / 3, .5], np.float32), tol=1e-7, what="right")


def test_kat_isolated_node_degree_zero(oracle):
    """deg 0 -> pow(0,-1/2)=inf -> replaced by 0 (gcn.py:23-29), not NaN."""
    ei = np.array([[0, 1], [1, 0]], np.int32)
    _, w = oracle.gcn_norm_adj(ei, None, 3, add_self_loop=False)
    assert np.isfinite(w).all() and np.allclose(w, [1, 1])
    out = oracle.gcn(np.eye(3, dtype=np.float32), ei, None, None, add_self_loop=False)
    assert np.array_equal(out[2], np.zeros(3, np.float32))


def test_kat_segment_softmax(oracle):
    s = np.array([0.0, np.log(3.0), 5.0, 1.0, 1.0], np.float32)
    ids = np.array([0, 0, 2, 1, 1], np.int32)
    got = oracle.segment_softmax(s, ids, 4)
    assert_parity(got, np.array([0.25, 0.75, 1.0, 0.5, 0.5], np.float32), tol=1e-6, what="softmax KAT")


def test_kat_gat_uniform_attention(oracle):
    """Zero Q/K kernels -> all scores 0 -> attention is the plain mean over {in-neighbours + self}."""
    n, f = 6, 3
    ei = np.array([[0, 0, 1, 2, 2, 2], [1, 2, 0, 3, 4, 5]], np.int32)
    rng = np.random.Generator(np.random.PCG64(0))
    x = rng.standard_normal((n, f), dtype=np.float32)
    z = np.zeros((f, 4), np.float32)
    out = oracle.gat(x, ei, z, np.zeros(4, np.float32), "relu", z, np.zeros(4, np.float32), "relu",
                     np.eye(f, dtype=np.float32), None, None, num_heads=1)
    exp = np.stack([(x[1] + x[2] + x[0]) / 3, (x[0] + x[1]) / 2, (x[3] + x[4] + x[5] + x[2]) / 4, x[3], x[4], x[5]])
    assert_parity(out, exp, tol=1e-6, what="uniform attention")


def test_kat_16_node_rings(oracle):
    """datasets/synthetic.py:47-66: two 8-rings (second is a 8-ring 8..15). Unweighted GCN on ones -> 1 (rows of the
    symmetric-normalised matrix of a regular graph sum to 1)."""
    ei = np.array([
        [0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 8],
        [1, 0, 2, 1, 3, 2, 0, 3, 5, 4, 6, 5, 7, 6, 4, 7, 9, 8, 10, 9, 11, 10, 12, 11, 13, 12, 14, 13, 15, 14, 8, 15]], np.int32)
    out = oracle.gcn(np.ones((16, 1), np.float32), ei, None, None)
    assert_parity(out, np.ones((16, 1), np.float32), tol=1e-6, what="regular graph")


# ------------------------------------------------------------------ (2) independent implementations
def _rand(oracle, n=300, e=4000, f=9, seed=0):
    ei = oracle.synthetic_edges(n, e, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    return rng.standard_normal((n, f), dtype=np.float32), ei, rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)


def test_cross_torch_cpu(oracle):
    x, ei, w = _rand(oracle)
    xt, row, col, wt = torch.from_numpy(x), torch.from_numpy(ei[0]).long(), torch.from_numpy(ei[1]).long(), torch.from_numpy(w)
    msg = xt[col] * wt[:, None]
    s = torch.zeros_like(xt).index_add_(0, row, msg)
    assert_parity(oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater),
                  s.numpy(), what="sum vs torch.index_add_")
    mx = torch.full_like(xt, float(oracle.FLT_LOWEST)).scatter_reduce(0, row[:, None].expand_as(msg), msg, "amax")
    assert np.array_equal(oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.max_reducer,
                                                     oracle.identity_updater), mx.numpy())
    mean = torch.zeros_like(xt).scatter_reduce(0, row[:, None].expand_as(msg), msg, "mean", include_self=False)
    assert_parity(oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.mean_reducer, oracle.identity_updater),
                  mean.numpy(), what="mean vs torch.scatter_reduce")


def _clib():
    path = os.path.join(ROOT, "oracle", "libtfg_oracle.so")
    if not os.path.exists(path):      # a fresh checkout: build the checker (gcc, a second) instead of skipping
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(path)


@pytest.mark.parametrize("op", [0, 1, 2])
@pytest.mark.parametrize("threads", [1, 3])
def test_cross_c_restatement(oracle, op, threads):
    lib = _clib()
    x, ei, w = _rand(oracle, seed=op)
    n, f = x.shape
    row, col = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    out = np.empty_like(x)
    P = ctypes.c_void_p
    rc = lib.tfgo_aggregate_coo_f32(P(x.ctypes.data), ctypes.c_int64(f), P(row.ctypes.data), P(col.ctypes.data),
                                    P(w.ctypes.data), ctypes.c_int64(ei.shape[1]), ctypes.c_int64(n), ctypes.c_int64(n),
                                    ctypes.c_int64(f), ctypes.c_int(op), P(out.ctypes.data), ctypes.c_int64(f),
                                    ctypes.c_int(threads))
    assert rc == 0
    red = [oracle.sum_reducer, oracle.mean_reducer, oracle.max_reducer][op]
    assert_parity(out, oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, red, oracle.identity_updater),
                  what="C COO op {}".format(op))
    # CSR variant used by bench.py's cpu_baseline
    order = np.argsort(row, kind="stable")
    rp = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(row, minlength=n), out=rp[1:])
    c2, w2 = np.ascontiguousarray(col[order]), np.ascontiguousarray(w[order])
    out2 = np.empty_like(x)
    rc = lib.tfgo_aggregate_csr_f32(P(x.ctypes.data), ctypes.c_int64(f), P(rp.ctypes.data), P(c2.ctypes.data),
                                    P(w2.ctypes.data), ctypes.c_int64(n), ctypes.c_int64(f), ctypes.c_int(op),
                                    P(out2.ctypes.data), ctypes.c_int64(f), ctypes.c_int(threads))
    assert rc == 0 and np.array_equal(out, out2)
    bad = row.copy()
    bad[0] = n
    assert lib.tfgo_aggregate_coo_f32(P(x.ctypes.data), ctypes.c_int64(f), P(bad.ctypes.data), P(col.ctypes.data),
                                      P(w.ctypes.data), ctypes.c_int64(ei.shape[1]), ctypes.c_int64(n),
                                      ctypes.c_int64(n), ctypes.c_int64(f), ctypes.c_int(0), P(out.ctypes.data),
                                      ctypes.c_int64(f), ctypes.c_int(1)) == 1


def test_cross_c_softmax(oracle):
    lib = _clib()
    rng = np.random.Generator(np.random.PCG64(3))
    ids = rng.integers(0, 40, size=2000, dtype=np.int32)
    s = (rng.standard_normal((2000, 4)) * 3).astype(np.float32)
    out = np.empty_like(s)
    P = ctypes.c_void_p
    assert lib.tfgo_segment_softmax_f32(P(s.ctypes.data), P(ids.ctypes.data), ctypes.c_int64(2000), ctypes.c_int64(4),
                                        ctypes.c_int64(40), P(out.ctypes.data)) == 0
    ref = np.stack([oracle.segment_softmax(s[:, h], ids, 40) for h in range(4)], axis=1)
    assert_parity(out, ref, what="C softmax")


def test_cross_dense_gcn_and_gat(oracle):
    """Dense-matrix formulations (no segment ops at all) of GCN and GAT."""
    x, ei, w = _rand(oracle, n=60, e=500, f=5, seed=7)
    n = 60
    A = np.zeros((n, n))
    np.add.at(A, (ei[0], ei[1]), w.astype(np.float64))
    A_hat = A + np.eye(n)
    d = A_hat.sum(1)
    dense = (A_hat / np.sqrt(d)[:, None] / np.sqrt(d)[None, :]) @ x.astype(np.float64)
    assert_parity(oracle.gcn(x, ei, w, None), dense, what="dense GCN")
    rng = np.random.Generator(np.random.PCG64(8))
    wq, wk, wv = [oracle.glorot_uniform(rng, 5, 6) for _ in range(3)]
    H, dh = 2, 3
    Q, K, V = np.maximum(x @ wq, 0).astype(np.float64), np.maximum(x @ wk, 0).astype(np.float64), (x @ wv).astype(np.float64)
    cnt = np.zeros((n, n))
    np.add.at(cnt, (ei[0], ei[1]), 1.0)
    cnt += np.eye(n)                      # multiplicity of each (row, col) pair incl. the appended self-loop
    outs = []
    for h in range(H):
        S = Q[:, h * dh:(h + 1) * dh] @ K[:, h * dh:(h + 1) * dh].T / np.sqrt(dh)
        P_ = cnt * np.exp(S - np.where(cnt > 0, S, -np.inf).max(1, keepdims=True))
        outs.append((P_ / P_.sum(1, keepdims=True)) @ V[:, h * dh:(h + 1) * dh])
    dense = np.concatenate(outs, axis=1)
    got = oracle.gat(x, ei, wq, np.zeros(6, np.float32), "relu", wk, np.zeros(6, np.float32), "relu", wv, None, None,
                     num_heads=2)
    assert_parity(got, dense, what="dense GAT")


# ------------------------------------------------------------------ (3) properties
def test_properties(oracle):
    x, ei, w = _rand(oracle, seed=11)
    agg = lambda xx, e, ww: oracle.aggregate_neighbors(xx, e, ww, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater)
    p = np.random.Generator(np.random.PCG64(1)).permutation(ei.shape[1])
    assert_parity(agg(x, ei, w), agg(x, ei[:, p], w[p]), tol=1e-6, what="edge-order invariance")
    assert_parity(agg(2 * x, ei, w), 2 * agg(x, ei, w), tol=1e-6, what="linearity in x")
    assert_parity(agg(x, np.concatenate([ei, ei], 1), np.concatenate([w, w])), 2 * agg(x, ei, w), tol=1e-6,
                  what="duplicate edges sum")
    assert np.array_equal(oracle.aggregate_neighbors(x, np.zeros((0,), np.int32), None), x)   # map_reduce.py:57
    with pytest.raises(ValueError):
        oracle.unsorted_segment_sum(x[:3], np.array([0, 1, 9]), 3)
    with pytest.raises(Exception):
        oracle.max_pool_graph_sage(x, ei, None, None, None, None)      # gcn_mapper(None) (graph_sage.py:260)
    # fp32-accumulated mode stays within the parity band of the float64 mode at these degrees
    assert_parity(oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater,
                                             acc=np.float32), agg(x, ei, w), what="fp32 vs fp64 accumulation")


# ------------------------------------------------------------------ (4) committed fixtures
def test_topk_pool_known_answers(oracle):
    """Hand-derived: source 0 holds items 1 (.5), 3 (.5), 6 (.7); source 2 holds 0 (.1), 2 (.9), 4 (.3); source 5 holds
    5.  Sources ascending, scores descending, the lower position wins the .5 tie (nn/pool/topk_pool.py:59)."""
    src = np.array([2, 0, 2, 0, 2, 5, 0])
    sc = np.array([.1, .5, .9, .5, .3, .2, .7], dtype=np.float32)
    assert oracle.topk_pool(src, sc, k=2).tolist() == [6, 1, 2, 4, 5]
    assert oracle.topk_pool(src, sc, k=1).tolist() == [6, 2, 5]
    assert oracle.topk_pool(src, sc, ratio=0.5).tolist() == [6, 1, 2, 4, 5]          # ceil(1.5) = 2, ceil(.5) = 1
    assert oracle.topk_pool(src, sc, ratio=1.0).tolist() == [6, 1, 3, 2, 4, 0, 5]
    assert oracle.topk_pool(src, sc, k=0).tolist() == []
    with pytest.raises(Exception):
        oracle.topk_pool(src, sc)
    with pytest.raises(Exception):
        oracle.topk_pool(src, sc, k=1, ratio=0.5)


def test_gcn_on_a_cut_out_with_whole_graph_degrees_equals_the_whole_graph_rows(oracle):
    """oracle.gcn(..., row_deg=...) — the form tests/test_gpu_fullsize.py runs on sub-problems cut out of a 123 M-edge graph:
    sampled destination rows + their in-edges + the row sums of the WHOLE graph for the nodes involved reproduce the whole
    graph's output rows (bitwise: same edges in the same order per row, same degrees)."""
    rng = np.random.Generator(np.random.PCG64(90))
    n, f, u = 500, 9, 6
    ei = oracle.synthetic_edges(n, 6000, seed=90)
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    k, b = oracle.glorot_uniform(rng, f, u), rng.standard_normal(u).astype(np.float32)
    whole = oracle.gcn(x, ei, w, k, b, "relu")
    rows = np.sort(rng.permutation(n)[:40])
    keep = np.isin(ei[0], rows)
    nodes = np.unique(np.concatenate([rows, ei[1][keep]]))
    sub = np.stack([np.searchsorted(nodes, ei[0][keep]), np.searchsorted(nodes, ei[1][keep])]).astype(np.int32)
    deg = oracle.unsorted_segment_sum(w.astype(np.float64), ei[0], n) + 1.0          # rows of A + I (gcn.py:77,80)
    part = oracle.gcn(x[nodes], sub, w[keep], k, b, "relu", row_deg=deg[nodes])
    assert np.array_equal(part[np.searchsorted(nodes, rows)], whole[rows])


def test_kat_sorted_segment_ops_and_pad(oracle):
    """tf.math.segment_* documented example (c = [[1,2,3,4],[4,3,2,1],[5,6,7,8]], ids [0,0,1]) and the rule that an id which
    does not occur yields 0; segment_op_with_pad (nn/kernel/segment.py:5-23) sorts, reduces and appends zero rows."""
    import functools
    c = np.array([[1, 2, 3, 4], [4, 3, 2, 1], [5, 6, 7, 8]], dtype=np.float32)
    ids = np.array([0, 0, 1])
    assert np.array_equal(oracle.sorted_segment("sum", c, ids), [[5, 5, 5, 5], [5, 6, 7, 8]])
    assert np.array_equal(oracle.sorted_segment("mean", c, ids), [[2.5, 2.5, 2.5, 2.5], [5, 6, 7, 8]])
    assert np.array_equal(oracle.sorted_segment("max", c, ids), [[4, 3, 3, 4], [5, 6, 7, 8]])
    assert np.array_equal(oracle.sorted_segment("min", c, ids), [[1, 2, 2, 1], [5, 6, 7, 8]])
    gap = oracle.sorted_segment("max", -c, np.array([0, 2, 2]))                 # id 1 never occurs: 0, not lowest
    assert np.array_equal(gap, [[-1, -2, -3, -4], [0, 0, 0, 0], [-4, -3, -2, -1]])
    with pytest.raises(ValueError):
        oracle.sorted_segment("sum", c, np.array([1, 0, 1]))
    out = oracle.segment_op_with_pad(functools.partial(oracle.sorted_segment, "max"), -c, np.array([2, 0, 2]), 5)
    assert np.array_equal(out, [[-4, -3, -2, -1], [0, 0, 0, 0], [-1, -2, -3, -4], [0, 0, 0, 0], [0, 0, 0, 0]])
