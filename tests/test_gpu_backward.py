# coding=utf-8
"""Backward pass: gradients of the HIP kernels vs torch autograd over a float64 dense/index
restatement of the same maths (the reference differentiates these ops with tf.GradientTape, demo/demo_gcn.py:68-77)."""
import numpy as np
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _graph(oracle, n=300, e=3000, f=12, seed=0):
    ei = oracle.synthetic_edges(n, e, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = rng.standard_normal((n, f)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
    return x, ei, w, rng


def _ref_aggregate(x, ei, w, op, n):
    row, col = torch.from_numpy(ei[0]).long(), torch.from_numpy(ei[1]).long()
    msg = x[col] * w[:, None] if w is not None else x[col]
    if op == "sum":
        return torch.zeros(n, x.shape[1], dtype=x.dtype).index_add(0, row, msg)
    if op == "mean":
        s = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add(0, row, msg)
        cnt = torch.bincount(row, minlength=n).clamp(min=1).to(x.dtype)
        return s / cnt[:, None]
    out = torch.full((n, x.shape[1]), -3.4028234663852886e38, dtype=x.dtype)
    return out.scatter_reduce(0, row[:, None].expand_as(msg), msg, "amax")


@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("weighted", [True, False])
def test_aggregate_grad_x_and_w(tfg, oracle, op, weighted):
    x, ei, w, rng = _graph(oracle, seed=3)
    n = x.shape[0]
    gout = rng.standard_normal(x.shape).astype(np.float32)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    wt = torch.tensor(w, device="cuda", requires_grad=weighted) if weighted else None
    red = getattr(tfg.nn, op + "_reducer")
    mapper = tfg.nn.gcn_mapper if weighted else tfg.nn.identity_mapper
    out = tfg.nn.aggregate_neighbors(xt, ei, wt, mapper, red, tfg.nn.sum_updater)
    out.backward(torch.tensor(gout, device="cuda"))
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = torch.tensor(w, dtype=torch.float64, requires_grad=True) if weighted else None
    ref = xr + _ref_aggregate(xr, ei, wr, op, n)
    ref.backward(torch.tensor(gout, dtype=torch.float64))
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what="forward " + op)
    assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=2e-5, what="d/dx " + op)
    if weighted:      # max included: d/dw flows to the edges that attain the row maximum (tfgx_segment_max_backward_w_f32)
        assert_parity(wt.grad.cpu().numpy(), wr.grad.numpy(), tol=2e-5, what="d/dw " + op)


@pytest.mark.parametrize("f", [4, 16, 24, 37, 64, 100, 128, 200, 256, 300, 512, 516])
def test_sddmm_widths(tfg, oracle, f):
    """tfgx_sddmm_f32 out[i] = <a[row(i)], b[col[i]]> on every dispatch (tuned float4 kernel 16 <= F <= 512, F % 4 == 0;
    scalar kernel otherwise), rows with 0 / 1 / 8 / 9 / many edges."""
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.plan import CsrPlan
    rng = np.random.Generator(np.random.PCG64(f))
    n = 600
    ei = oracle.synthetic_edges(n, 9000, seed=f)
    ei = ei[:, ei[0] != 3]                                              # row 3 empty
    extra = np.stack([np.full(700, 5, np.int32), rng.integers(0, n, 700).astype(np.int32)])   # a long row
    ei = np.concatenate([ei, extra], axis=1)
    a = rng.standard_normal((n, f), dtype=np.float32)
    b = rng.standard_normal((n, f), dtype=np.float32)
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    at, bt = L.as_f32(a), L.as_f32(b)
    out = torch.empty(plan.num_edges, dtype=torch.float32, device="cuda")
    lib = L.require_gpu()
    L.check(lib.tfgx_sddmm_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), n, L.ptr(at), f, L.ptr(bt), f, f, L.ptr(out),
                               L.stream_ptr()), "tfgx_sddmm_f32")
    rp, col = plan.row_ptr.cpu().numpy(), plan.col.cpu().numpy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    ref = np.einsum("ij,ij->i", a[rows].astype(np.float64), b[col].astype(np.float64))
    assert_parity(out.cpu().numpy(), ref, tol=1e-5 * np.sqrt(f), what="sddmm F={}".format(f))


@pytest.mark.parametrize("f,weighted", [(100, True), (64, False), (16, True), (37, True), (260, False)])
def test_max_with_count_one_pass_equals_forward_plus_count(tfg, oracle, f, weighted):
    """tfgx_segment_max_with_count_f32 (training forward): bit-identical maxima to the plain forward and the same tie
    counts as the separate count pass — empty rows, duplicated sources (ties), aligned and unaligned widths."""
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.plan import CsrPlan, segment_reduce
    rng = np.random.Generator(np.random.PCG64(f))
    n = 700
    ei = oracle.synthetic_edges(n, 9000, seed=f)
    ei = ei[:, ei[0] != 9]
    ei = np.concatenate([ei, ei[:, :3000]], axis=1)                     # duplicated edges: guaranteed ties
    x = np.round(rng.standard_normal((n, f)).astype(np.float32) * 2) / 2   # quantised: more ties
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    xd = L.as_f32(x)
    w = (torch.randint(1, 3, (plan.num_edges,), device="cuda").float() * 0.5) if weighted else None
    lib = L.require_gpu()
    out = torch.empty((n, f), device="cuda")
    cnt = torch.empty((n, f), device="cuda")
    L.check(lib.tfgx_segment_max_with_count_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w), n, L.ptr(xd), f, f,
                                                L.ptr(out), f, L.ptr(cnt), f, L.stream_ptr()), "max_with_count")
    ref_out = segment_reduce(plan, xd, L.MAX, w_csr=w)
    ref_cnt = torch.empty_like(ref_out)
    L.check(lib.tfgx_segment_max_count_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w), n, L.ptr(xd), f, f,
                                           L.ptr(ref_out), f, L.ptr(ref_cnt), f, L.stream_ptr()), "max_count")
    assert torch.equal(out, ref_out) and torch.equal(cnt, ref_cnt)
    assert float(cnt.max()) >= 2 and float(cnt[9].abs().max()) == 0 and float(out[9].max()) == -3.4028234663852886e38


def test_max_grad_ties_split_evenly(tfg):
    """TF's unsorted_segment_max gradient divides by the number of tied maxima."""
    ei = np.array([[0, 0, 0, 1], [1, 2, 3, 3]], np.int32)
    x = np.array([[0.0], [5.0], [5.0], [1.0]], np.float32)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = tfg.nn.aggregate_neighbors(xt, ei, None, tfg.nn.identity_mapper, tfg.nn.max_reducer, tfg.nn.identity_updater)
    out[:2].sum().backward()
    assert np.allclose(xt.grad.cpu().numpy()[:, 0], [0.0, 0.5, 0.5, 1.0])


@pytest.mark.parametrize("cfg", [dict(), dict(renorm=False), dict(norm="left")])
@pytest.mark.parametrize("units", [8, 24])
def test_gcn_layer_grads(tfg, oracle, cfg, units):
    x, ei, w, rng = _graph(oracle, seed=5)
    n, f = x.shape
    layer = tfg.layers.GCN(units, activation=tfg.relu, **cfg)
    layer._maybe_build([x])
    layer.set_weights(kernel=oracle.glorot_uniform(rng, f, units), bias=(rng.standard_normal(units) * 0.1).astype(np.float32))
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = layer([xt, ei, w], cache={})
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    out.backward(gout)
    # float64 reference: dense normalised adjacency from the oracle
    idx, nw = oracle.gcn_norm_adj(ei, w, n, **cfg)
    A = torch.zeros(n, n, dtype=torch.float64).index_put((torch.from_numpy(idx[0]).long(), torch.from_numpy(idx[1]).long()),
                                                         torch.from_numpy(nw).double(), accumulate=True)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    kr = layer.kernel.detach().double().cpu().requires_grad_(True)
    br = layer.bias.detach().double().cpu().requires_grad_(True)
    ref = torch.relu(A @ (xr @ kr) + br)
    ref.backward(gout.double().cpu())
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what="gcn forward")
    assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=2e-5, what="gcn d/dx")
    assert_parity(layer.kernel.grad.cpu().numpy(), kr.grad.numpy(), tol=1e-4, what="gcn d/dkernel")
    assert_parity(layer.bias.grad.cpu().numpy(), br.grad.numpy(), tol=1e-4, what="gcn d/dbias")


def _ref_gat(x, ei, wq, bq, wk, bk, wv, b, H, n):
    ar = np.arange(n, dtype=np.int64)
    row = torch.from_numpy(np.concatenate([ei[0].astype(np.int64), ar]))
    col = torch.from_numpy(np.concatenate([ei[1].astype(np.int64), ar]))
    Q, K, V = torch.relu(x @ wq + bq), torch.relu(x @ wk + bk), x @ wv
    d, dv = Q.shape[1] // H, V.shape[1] // H
    outs = []
    for h in range(H):
        s = (Q[row, h * d:(h + 1) * d] * K[col, h * d:(h + 1) * d]).sum(-1) / np.sqrt(d)
        m = torch.full((n,), -1e30, dtype=x.dtype).scatter_reduce(0, row, s, "amax")
        p = torch.exp(s - m[row].detach())
        den = torch.zeros(n, dtype=x.dtype).index_add(0, row, p) + 1e-8
        a = p / den[row]
        outs.append(torch.zeros(n, dv, dtype=x.dtype).index_add(0, row, a[:, None] * V[col, h * dv:(h + 1) * dv]))
    return torch.relu(torch.cat(outs, 1) + b)


def _keep_mask_host(seed, n_items, rate):
    """numpy restatement of drop_hash / drop_scale (tf_geometric_amd/csrc/tfgx_common.h) — test infrastructure."""
    x = np.arange(n_items, dtype=np.uint64).astype(np.uint32) ^ np.uint32(seed & 0xFFFFFFFF)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x85ebca6b)).astype(np.uint32)
    x ^= np.uint32(seed >> 32)
    x ^= x >> np.uint32(13)
    x = (x.astype(np.uint64) * np.uint64(0xc2b2ae35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return (x >> np.uint32(8)) >= np.uint32(int(np.float32(rate) * np.float32(16777216.0)))


@pytest.mark.parametrize("heads,att,units,rate,hubs", [(1, 4, 8, 0.5, False), (4, 8, 16, 0.6, False), (2, 32, 8, 0.25, False),
                                                       (2, 6, 10, 0.4, False), (4, 8, 16, 0.5, True)])
def test_gat_attention_dropout_forward_and_grads(tfg, oracle, heads, att, units, rate, hubs):
    """Attention dropout (SparseMatrix.dropout after segment_softmax, gat.py:85): out = sum_e a_e keep_e/(1-rate) V.
    The mask is regenerated on the host from (seed, CSR position, head); forward and dQ, dK, dV are compared with torch
    autograd over a float64 restatement that uses that mask.  Covers the tuned and the one-lane-per-row backward."""
    import tf_geometric_amd.autograd as AG
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.plan import CsrPlan
    x, ei, w, rng = _graph(oracle, n=300, e=4000, f=5, seed=heads + att)
    n = x.shape[0]
    import tf_geometric_amd.plan as P
    old_policy = (P.HUB_THRESHOLD, P.HUB_CHUNK)
    if hubs:    # a 3000-in-edge destination and a 2500-out-edge source, a low threshold: the backward passes run chunk-wise
        # (and in degree order) while the keep mask still comes from the edges' absolute CSR positions
        ei = np.concatenate([np.stack([np.full(3000, 9, np.int32), rng.integers(0, n, 3000, dtype=np.int32)]), ei,
                             np.stack([rng.integers(0, n, 2500, dtype=np.int32), np.full(2500, 4, np.int32)])], axis=1)
        P.HUB_THRESHOLD, P.HUB_CHUNK = 64, 48
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    if hubs:
        assert plan.hub_info() is not None and plan.transposed().hub_info() is not None and plan.row_order() is not None
    P.HUB_THRESHOLD, P.HUB_CHUNK = old_policy
    E = plan.num_edges
    seed = (0x1234ABCD << 32) | (77 + heads)
    lib = L.require_gpu()
    keep = _keep_mask_host(seed, (E + n) * heads, rate)
    for item in (0, 1, 17, (E + n) * heads - 1, E * heads + 3):
        assert bool(keep[item]) == bool(lib.tfgx_dropout_keep(seed, int(item), float(rate)))
    assert abs(keep.mean() - (1.0 - rate)) < 0.02
    keep = keep.reshape(E + n, heads)
    Q = rng.standard_normal((n, att)).astype(np.float32)
    K = rng.standard_normal((n, att)).astype(np.float32)
    V = rng.standard_normal((n, units)).astype(np.float32)
    gout = rng.standard_normal((n, units)).astype(np.float32)
    t = {k: torch.tensor(v, device="cuda", requires_grad=True) for k, v in dict(Q=Q, K=K, V=V).items()}
    out = AG.gat_attention(plan, t["Q"], t["K"], t["V"], heads, drop_rate=rate, drop_seed=seed)
    out.backward(torch.tensor(gout, device="cuda"))
    # float64 reference in CSR order, self-loops appended (positions E .. E+n-1)
    rp = plan.row_ptr.cpu().numpy()
    ar = np.arange(n, dtype=np.int64)
    row = torch.from_numpy(np.concatenate([np.repeat(ar, np.diff(rp)), ar]))
    col = torch.from_numpy(np.concatenate([plan.col.cpu().numpy().astype(np.int64), ar]))
    r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in dict(Q=Q, K=K, V=V).items()}
    d, dv = att // heads, units // heads
    outs = []
    for h in range(heads):
        s = (r["Q"][row, h * d:(h + 1) * d] * r["K"][col, h * d:(h + 1) * d]).sum(-1) / np.sqrt(d)
        m = torch.full((n,), -1e30, dtype=torch.float64).scatter_reduce(0, row, s, "amax")
        p = torch.exp(s - m[row].detach())
        den = torch.zeros(n, dtype=torch.float64).index_add(0, row, p) + 1e-8
        a = p / den[row] * torch.from_numpy(keep[:, h].astype(np.float64)) / (1.0 - rate)
        outs.append(torch.zeros(n, dv, dtype=torch.float64).index_add(0, row, a[:, None] * r["V"][col, h * dv:(h + 1) * dv]))
    ref = torch.cat(outs, 1)
    ref.backward(torch.tensor(gout, dtype=torch.float64))
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), tol=2e-5, what="dropout forward")
    for k in ("Q", "K", "V"):
        assert_parity(t[k].grad.cpu().numpy(), r[k].grad.numpy(), tol=5e-5, what="dropout d/d" + k)
    # rate 0 through the same entry == the plain kernel
    plain = AG.gat_attention(plan, t["Q"].detach(), t["K"].detach(), t["V"].detach(), heads)
    assert not torch.equal(plain, out.detach())


def test_gat_layer_edge_dropout_training(tfg, oracle):
    """layers.GAT(edge_drop_rate=...) as demo/demo_gat.py:22 builds it: identity at inference, active under
    training=True, repeatable under torch.manual_seed, unbiased on average, differentiable."""
    x, ei, w, rng = _graph(oracle, n=400, e=6000, f=12, seed=5)
    layer = tfg.layers.GAT(16, attention_units=8, activation=tfg.relu, num_heads=4, edge_drop_rate=0.6)
    ev = layer([x, ei], training=False)
    layer0 = tfg.layers.GAT(16, attention_units=8, activation=tfg.relu, num_heads=4)
    layer0._maybe_build([x])
    layer0.set_weights(**{k: getattr(layer, k).detach().cpu().numpy() for k in
                          ("query_kernel", "key_kernel", "kernel", "query_bias", "key_bias", "bias")})
    assert torch.equal(ev, layer0([x, ei]))
    torch.manual_seed(11)
    t1 = layer([x, ei], training=True)
    torch.manual_seed(11)
    t2 = layer([x, ei], training=True)
    t3 = layer([x, ei], training=True)
    assert torch.equal(t1, t2) and not torch.equal(t1, t3) and not torch.equal(t1, ev)
    layer.activation = None
    ev_lin = layer([x, ei], training=False)
    acc = torch.zeros_like(ev_lin)
    for _ in range(200):
        acc += layer([x, ei], training=True)
    err = float((acc / 200 - ev_lin).abs().mean() / ev_lin.abs().mean())
    assert err < 0.1, err                                   # E[dropout(a)] = a
    layer.trainable(True)
    out = layer([torch.tensor(x, device="cuda", requires_grad=True), ei], training=True)
    out.square().sum().backward()
    assert layer.kernel.grad is not None and float(layer.kernel.grad.abs().sum()) > 0
    assert float(layer.query_kernel.grad.abs().sum()) > 0


def test_gcn_edge_dropout_training(tfg, oracle):
    """GCN(edge_drop_rate=...) (gcn.py:262: SparseMatrix.dropout on the normalised adjacency): identity at inference,
    unbiased and repeatable in training, plan shared, gradients flow; SparseMatrix.dropout likewise."""
    x, ei, w, rng = _graph(oracle, n=500, e=8000, f=10, seed=8)
    layer = tfg.layers.GCN(6, edge_drop_rate=0.5)
    cache = {}
    ev = layer([x, ei, w], cache=cache, training=False)
    assert torch.equal(ev, layer([x, ei, w], cache=cache))
    torch.manual_seed(3)
    t1 = layer([x, ei, w], cache=cache, training=True)
    torch.manual_seed(3)
    t2 = layer([x, ei, w], cache=cache, training=True)
    assert torch.equal(t1, t2) and not torch.equal(t1, ev)
    acc = torch.zeros_like(ev)
    for _ in range(300):
        acc += layer([x, ei, w], cache=cache, training=True)
    assert float((acc / 300 - ev).abs().mean() / ev.abs().mean()) < 0.1
    layer.trainable(True)
    out = layer([torch.tensor(x, device="cuda", requires_grad=True), ei, w], cache=cache, training=True)
    out.square().sum().backward()
    assert float(layer.kernel.grad.abs().sum()) > 0
    adj = tfg.SparseMatrix(ei, w, [500, 500])
    _ = adj.plan                                            # built once; the dropped copy shares it
    assert adj.dropout(0.3, training=False) is adj
    torch.manual_seed(1)
    d = adj.dropout(0.3, training=True)
    kept = d.value != 0
    assert abs(float(kept.float().mean()) - 0.7) < 0.03 and d.plan is adj.plan
    assert torch.allclose(d.value[kept], adj.value[kept] / 0.7, rtol=1e-6)


@pytest.mark.parametrize("heads,att,units", [(1, 4, 6), (4, 8, 16), (2, 6, 10), (1, 1, 41), (1, 2, 7), (2, 64, 8), (4, 12, 20),
                                             (2, 2, 82), (3, 9, 9), (1, 4, 300)])
def test_gat_layer_grads(tfg, oracle, heads, att, units):
    x, ei, w, rng = _graph(oracle, n=200, e=2500, f=9, seed=7)
    n, f = x.shape
    layer = tfg.layers.GAT(units, attention_units=att, activation=tfg.relu, num_heads=heads)
    layer._maybe_build([x])
    ws = dict(query_kernel=oracle.glorot_uniform(rng, f, att), key_kernel=oracle.glorot_uniform(rng, f, att),
              kernel=oracle.glorot_uniform(rng, f, units), query_bias=(rng.standard_normal(att) * 0.3).astype(np.float32),
              key_bias=(rng.standard_normal(att) * 0.3).astype(np.float32), bias=(rng.standard_normal(units) * 0.1).astype(np.float32))
    layer.set_weights(**ws)
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = layer([xt, ei])
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    out.backward(gout)
    r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = _ref_gat(xr, ei, r["query_kernel"], r["query_bias"], r["key_kernel"], r["key_bias"], r["kernel"], r["bias"], heads, n)
    ref.backward(gout.double().cpu())
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what="gat forward")
    assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=5e-5, what="gat d/dx")
    for k in ws:
        assert_parity(getattr(layer, k).grad.cpu().numpy(), r[k].grad.numpy(), tol=2e-4, what="gat d/d" + k)


@pytest.mark.parametrize("skewed", [False, True])
@pytest.mark.parametrize("blocks", [None, 5])
@pytest.mark.parametrize("drop", [0.0, 0.4])
def test_gat_query_gradient_from_the_forward_sums(tfg, oracle, blocks, drop, skewed):
    """One attention unit per head (the demo's literal layer): the training forward accumulates T = sum a k V and S = sum a k,
    dQ = (<dO, T> - D S) / scale per row — against the destination pass (the same inputs with the route switched off: the
    forward must be bit-identical, dQ equal to rounding) and against float64 autograd."""
    from tf_geometric_amd.nn.conv import gat as G
    from tf_geometric_amd import autograd as AG
    from tf_geometric_amd.plan import CsrPlan
    if blocks is not None and drop > 0.0:
        pytest.skip("source blocks run without attention dropout")
    if skewed and (drop == 0.0 or blocks is not None):
        pytest.skip("long rows run inline (degree-ordered walk) only under attention dropout; without it they are hub chunks")
    rng = np.random.default_rng(77)
    n, e, H, dv = 500, 30000, 8, 8
    ei = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)]).astype(np.int32)
    if skewed:        # five destinations with 900 in-edges each: the plan walks its rows in degree order
        ei = np.concatenate([ei, np.stack([np.repeat(np.arange(5), 900), rng.integers(0, n, 4500)]).astype(np.int32)], axis=1)
    plan = CsrPlan.from_cache(ei, n, n, {})
    assert (plan.row_order() is not None) == skewed
    Qn, Kn, Vn = (rng.standard_normal((n, H)).astype(np.float32) * 1.5, rng.standard_normal((n, H)).astype(np.float32) * 1.5,
                  rng.standard_normal((n, H * dv)).astype(np.float32))
    gout = torch.tensor(rng.standard_normal((n, H * dv)).astype(np.float32), device="cuda")

    def run(on):
        Q, K, V = (torch.tensor(t, device="cuda", requires_grad=True) for t in (Qn, Kn, Vn))
        before = G.SOURCE_BLOCK_STATS.get("query_sum_backwards", 0)
        G.QUERY_GRAD_SUMS, G.SOURCE_BLOCKS = on, blocks
        try:
            out = AG.gat_attention(plan, Q, K, V, H, drop_rate=drop, drop_seed=1234)
            out.backward(gout)
        finally:
            G.QUERY_GRAD_SUMS, G.SOURCE_BLOCKS = True, None
        assert G.SOURCE_BLOCK_STATS.get("query_sum_backwards", 0) == before + (1 if on else 0)
        return out.detach(), Q.grad, K.grad, V.grad

    o1, q1, k1, v1 = run(True)
    o0, q0, k0, v0 = run(False)
    assert torch.equal(o1, o0) and torch.equal(k1, k0) and torch.equal(v1, v0)
    scale = float(q0.abs().max())
    # two float32 evaluations of the same sums in different association (measured beside float64, tools/r06/diag_query_sums.py:
    # 1.6e-5 / 1.1e-5 off at this shape, whose softmaxes are peaked — scores of std 2 — and whose largest |dQ| is 5)
    assert float((q1 - q0).abs().max()) <= 5e-6 * max(scale, 1.0), (float((q1 - q0).abs().max()), scale)
    if drop == 0.0:      # float64 autograd of the reference's formulation (gat.py:73-89 with the self-loop appended)
        Q, K, V = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (Qn, Kn, Vn))
        row = torch.cat([torch.tensor(ei[0]).long(), torch.arange(n)])
        col = torch.cat([torch.tensor(ei[1]).long(), torch.arange(n)])
        s = Q[row] * K[col]                                                     # d = 1: scale sqrt(1)
        mx = torch.full((n, H), -float("inf"), dtype=torch.float64).scatter_reduce(0, row[:, None].expand(-1, H), s.detach(), "amax")
        ex = torch.exp(s - mx[row])
        den = torch.zeros((n, H), dtype=torch.float64).index_add(0, row, ex) + 1e-8
        alpha = ex / den[row]
        out = torch.zeros((n, H, dv), dtype=torch.float64).index_add(0, row, alpha[:, :, None] * V[col].reshape(-1, H, dv))
        out.reshape(n, H * dv).backward(gout.double().cpu())
        assert_parity(o1.cpu().numpy(), out.detach().reshape(n, H * dv).numpy(), what="attention forward")
        assert_parity(q1.cpu().numpy(), Q.grad.numpy(), tol=2e-5, what="dQ from the forward sums")


def test_gat_layer_grads_in_source_blocks(tfg, oracle):
    """Training forward in chained source-block launches, destination pass (dQ) in source blocks and source pass (dK, dV) in
    destination blocks with the gradients accumulated block by block: outputs and every gradient against float64 autograd."""
    from tf_geometric_amd.nn.conv import gat as G
    x, ei, w, rng = _graph(oracle, n=300, e=20000, f=9, seed=31)
    n, f = x.shape
    heads, att, units = 4, 8, 16
    layer = tfg.layers.GAT(units, attention_units=att, activation=tfg.relu, num_heads=heads)
    layer._maybe_build([x])
    ws = dict(query_kernel=oracle.glorot_uniform(rng, f, att), key_kernel=oracle.glorot_uniform(rng, f, att),
              kernel=oracle.glorot_uniform(rng, f, units), query_bias=(rng.standard_normal(att) * 0.3).astype(np.float32),
              key_bias=(rng.standard_normal(att) * 0.3).astype(np.float32), bias=(rng.standard_normal(units) * 0.1).astype(np.float32))
    layer.set_weights(**ws)
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    before, bw_before = G.SOURCE_BLOCK_STATS["launches"], G.SOURCE_BLOCK_STATS.get("backward_launches", 0)
    G.SOURCE_BLOCKS = 5
    try:
        out = layer([xt, ei])
        gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
        out.backward(gout)
    finally:
        G.SOURCE_BLOCKS = None
    assert G.SOURCE_BLOCK_STATS["launches"] == before + 5 and G.SOURCE_BLOCK_STATS.get("backward_launches", 0) >= bw_before + 10
    r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ref = _ref_gat(xr, ei, r["query_kernel"], r["query_bias"], r["key_kernel"], r["key_bias"], r["kernel"], r["bias"], heads, n)
    ref.backward(gout.double().cpu())
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what="gat forward (source blocks)")
    assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=5e-5, what="gat d/dx (source blocks)")
    for k in ws:
        assert_parity(getattr(layer, k).grad.cpu().numpy(), r[k].grad.numpy(), tol=2e-4, what="gat d/d" + k)


@pytest.mark.parametrize("cls", ["MeanGraphSage", "SumGraphSage", "MaxPoolGraphSage", "MeanPoolGraphSage", "GCNGraphSage"])
def test_sage_layers_train_step_decreases_loss(tfg, oracle, cls):
    """Every GraphSAGE variant is differentiable end to end: a few SGD steps reduce a regression loss."""
    x, ei, w, rng = _graph(oracle, n=250, e=3000, f=10, seed=9)
    n = x.shape[0]
    ring = np.stack([np.arange(n, dtype=np.int32), np.roll(np.arange(n, dtype=np.int32), 1)])
    ei = np.concatenate([ei, ring], axis=1)
    w = np.concatenate([w, np.ones(n, np.float32)])
    layer = getattr(tfg.layers, cls)(8)
    target = torch.tensor(rng.standard_normal((n, 8)).astype(np.float32), device="cuda").abs()
    layer._maybe_build([x])
    layer.trainable(True)
    opt = torch.optim.SGD(layer.parameters(), lr=0.05)
    losses = []
    cache = {}
    for _ in range(12):
        opt.zero_grad()
        out = layer([x, ei, w], cache=cache)
        loss = ((out - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] * 0.995 and all(b <= a + 1e-6 for a, b in zip(losses, losses[1:])), losses
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())


@pytest.mark.parametrize("cls,f,units,concat", [("MeanGraphSage", 10, 32, True), ("MeanGraphSage", 40, 16, True),
                                                 ("SumGraphSage", 40, 16, True), ("MeanGraphSage", 40, 16, False),
                                                 ("SumGraphSage", 12, 24, False),
                                                 # odd widths: the halves start at unaligned columns (ku = 5, 7, 3)
                                                 ("MeanGraphSage", 23, 10, True), ("SumGraphSage", 5, 14, True),
                                                 ("MeanGraphSage", 9, 6, True), ("SumGraphSage", 101, 130, True)])
def test_mean_sum_sage_layer_grads_fused_epilogues(tfg, oracle, cls, f, units, concat):
    """The training route of mean / sum GraphSAGE: with concat both halves are written in place with bias + ReLU in the
    GEMM / aggregation epilogues (autograd._DualLinear when the reduction runs at the input width, autograd._SageNarrow
    when the neighbour projection runs first), without concat the un-fused operators — outputs and the gradients of x,
    both kernels and the bias vs float64 autograd over graph_sage.py:9-115's formula."""
    x, ei, w, rng = _graph(oracle, n=260, e=2600, f=f, seed=21)
    n = x.shape[0]
    ei = ei[:, ei[0] != 5]                                             # an empty row: mean divides by max(count, 1)
    w = w[:ei.shape[1]]
    layer = getattr(tfg.layers, cls)(units, activation=tfg.relu, concat=concat)
    layer._maybe_build([x])
    ku = units // 2 if concat else units
    ws = {"self_kernel": oracle.glorot_uniform(rng, f, ku), "neighbor_kernel": oracle.glorot_uniform(rng, f, ku),
          "bias": (rng.standard_normal(units) * 0.3).astype(np.float32)}
    layer.set_weights(**ws)
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = layer([xt, ei, w], cache={})
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    out.backward(gout)
    r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    red = _ref_aggregate(xr, ei, torch.tensor(w, dtype=torch.float64), "mean" if cls.startswith("Mean") else "sum", n)
    a, b = xr @ r["self_kernel"], red @ r["neighbor_kernel"]
    ref = torch.relu((torch.cat([a, b], 1) if concat else a + b) + r["bias"])
    ref.backward(gout.double().cpu())
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what=cls + " forward")
    assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=5e-5, what=cls + " d/dx")
    for k in ws:
        assert_parity(getattr(layer, k).grad.cpu().numpy(), r[k].grad.numpy(), tol=2e-4, what=cls + " d/d" + k)


@pytest.mark.parametrize("kind,f,units", [("gcn", 12, 24), ("gcn", 100, 256), ("gcn", 128, 256), ("MeanGraphSage", 12, 32),
                                           ("SumGraphSage", 100, 256), ("MeanGraphSage", 128, 512)])
@pytest.mark.parametrize("x_grad", [True, False])
def test_training_forward_takes_the_fused_launch(tfg, oracle, kind, f, units, x_grad):
    """The aggregate-then-project layers TRAIN through tfgx_aggregate_gemm_f32 too (autograd._AggregateProject /
    _SageWide): one forward launch, the aggregate written beside it for the weight gradient (FUSED_STATS says so), and
    output + every gradient vs float64 autograd over the reference's formula (gcn.py:272-288, graph_sage.py:34-58) —
    including F = 128 -> 256, where half of the kernel is read from global memory.  x_grad False = layer 0 (data input)."""
    from tf_geometric_amd import plan as P
    x, ei, w, rng = _graph(oracle, n=700, e=9000, f=f, seed=31)
    n = x.shape[0]
    ei = ei[:, ei[0] != 7]
    w = w[:ei.shape[1]]
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    xt = torch.tensor(x, device="cuda", requires_grad=x_grad)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    before = dict(P.FUSED_STATS)
    if kind == "gcn":
        layer = tfg.layers.GCN(units, activation=tfg.relu)
        layer._maybe_build([x])
        ws = {"kernel": oracle.glorot_uniform(rng, f, units), "bias": (rng.standard_normal(units) * 0.1).astype(np.float32)}
        layer.set_weights(**ws)
        layer.trainable(True)
        out = layer([xt, ei, w], cache={})
        r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
        idx, nw = oracle.gcn_norm_adj(ei, w, n)
        A = torch.zeros(n, n, dtype=torch.float64).index_put((torch.from_numpy(idx[0]).long(), torch.from_numpy(idx[1]).long()),
                                                             torch.from_numpy(nw).double(), accumulate=True)
        ref = torch.relu(A @ (xr @ r["kernel"]) + r["bias"])
    else:
        layer = getattr(tfg.layers, kind)(units, activation=tfg.relu, concat=True)
        layer._maybe_build([x])
        ku = units // 2
        ws = {"self_kernel": oracle.glorot_uniform(rng, f, ku), "neighbor_kernel": oracle.glorot_uniform(rng, f, ku),
              "bias": (rng.standard_normal(units) * 0.3).astype(np.float32)}
        layer.set_weights(**ws)
        layer.trainable(True)
        out = layer([xt, ei, w], cache={})
        r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
        red = _ref_aggregate(xr, ei, torch.tensor(w, dtype=torch.float64), "mean" if kind.startswith("Mean") else "sum", n)
        ref = torch.relu(torch.cat([xr @ r["self_kernel"], red @ r["neighbor_kernel"]], 1) + r["bias"])
    assert P.FUSED_STATS["launches"] == before["launches"] + 1
    assert P.FUSED_STATS["with_side_output"] == before["with_side_output"] + 1      # the kernel's gradient needs the aggregate
    out.backward(gout)
    ref.backward(gout.double().cpu())
    assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), what=kind + " fused training forward")
    if x_grad:
        assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=5e-5, what=kind + " d/dx")
    for k in ws:
        assert_parity(getattr(layer, k).grad.cpu().numpy(), r[k].grad.numpy(), tol=2e-4, what=kind + " d/d" + k)


def test_demo_gcn_trains_on_cora_shaped_graph(tfg):
    """examples/demo_gcn.py (counterpart of the reference's demo/demo_gcn.py): accuracy far above 1/7 chance."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import demo_gcn
    acc, _ = demo_gcn.main(steps=60, forward_iters=0, quiet=True)
    assert acc > 0.5, acc
    acc, _ = demo_gcn.main(steps=60, forward_iters=0, quiet=True, hipgraph=True)     # the whole step (dropout included) replayed
    assert acc > 0.5, acc


def test_demo_gat_trains_with_attention_dropout(tfg):
    """examples/demo_gat.py (counterpart of demo/demo_gat.py: 8-head GAT, edge_drop_rate 0.6 in both layers, weights
    made trainable BEFORE the lazy build): the loss falls and the accuracy ends far above 1/7 chance."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import demo_gat
    acc, loss = demo_gat.main(steps=80, quiet=True)
    assert acc > 0.5 and loss < 1.9, (acc, loss)


def test_demo_graph_sage_trains_with_neighbour_sampling(tfg):
    """examples/demo_graph_sage.py (counterpart of demo/demo_graph_sage.py): a fresh k = 25 / 10 neighbour sample and a
    fresh CSR plan per layer and step, MeanGraphSage x 2, multi-label loss — micro-F1 on an unseen graph improves."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import demo_graph_sage
    f1, loss = demo_graph_sage.main(epochs=4, quiet=True, num_train=3)
    assert f1 > 0.68 and loss < 0.62, (f1, loss)


def test_gradients_reach_through_readouts_sparse_matmul_and_generic_route(tfg):
    """readouts, SparseMatrix.matmul / segment_softmax, the generic aggregate_neighbors route and
    nn.segment_softmax must carry a grad_fn (the reference's TF ops are all differentiable) — checked against torch
    autograd over the same maths in float64."""
    import torch
    dev = "cuda"
    rng = np.random.Generator(np.random.PCG64(31))
    n, f, g_ = 120, 7, 9
    x64 = torch.tensor(rng.standard_normal((n, f)), device=dev, requires_grad=True)
    gid = torch.tensor(np.sort(rng.integers(0, g_, n)), device=dev)
    ei = torch.tensor(rng.integers(0, n, size=(2, 900)), device=dev)
    w64 = torch.tensor(rng.uniform(0.5, 1.5, 900), device=dev, requires_grad=True)
    proj = torch.tensor(rng.standard_normal((g_, f)), device=dev)

    def check(got_fn, ref_fn, inputs64, what, tol=2e-4):
        ins32 = [t.detach().float().requires_grad_(True) for t in inputs64]
        out = got_fn(*ins32)
        assert out.grad_fn is not None, what + ": output is detached"
        out.double().mul(ref_fn.weight(out)).sum().backward()
        ref = ref_fn(*inputs64)
        grads64 = torch.autograd.grad(ref.mul(ref_fn.weight(ref)).sum(), inputs64)
        for a, b in zip(ins32, grads64):
            assert a.grad is not None, what + ": no gradient"
            assert_parity(a.grad.cpu().numpy(), b.cpu().numpy(), tol=tol, what=what)

    class Ref(object):
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, *a):
            return self.fn(*a)

        @staticmethod
        def weight(o):
            return torch.linspace(0.5, 1.5, o.numel(), device=o.device, dtype=torch.float64).reshape(o.shape)

    cnt = torch.bincount(gid, minlength=g_).double().unsqueeze(1)
    seg_sum = lambda v, ids, m: torch.zeros((m,) + v.shape[1:], dtype=v.dtype, device=dev).index_add_(0, ids, v)   # noqa: E731
    check(lambda x: tfg.nn.sum_pool(x, gid.int(), g_), Ref(lambda x: seg_sum(x, gid, g_)), [x64], "sum_pool")
    check(lambda x: tfg.nn.mean_pool(x, gid.int(), g_), Ref(lambda x: seg_sum(x, gid, g_) / (cnt + 1e-8)), [x64], "mean_pool")
    mx = lambda x: torch.stack([x[gid == i].max(0).values for i in range(g_)])      # noqa: E731
    check(lambda x: tfg.nn.max_pool(x, gid.int(), g_), Ref(mx), [x64], "max_pool")
    check(lambda x: tfg.nn.min_pool(x, gid.int(), g_), Ref(lambda x: -mx(-x)), [x64], "min_pool")
    check(lambda x: tfg.layers.MeanPool()([x, gid.int(), g_]), Ref(lambda x: seg_sum(x, gid, g_) / (cnt + 1e-8)), [x64],
          "MeanPool layer")
    # SparseMatrix.matmul: gradients wrt the dense operand AND the stored values
    check(lambda x, w: tfg.SparseMatrix(ei.int(), w, [n, n]) @ x,
          Ref(lambda x, w: seg_sum(x[ei[1]] * w.unsqueeze(1), ei[0], n)), [x64, w64], "SparseMatrix @ x")
    # segment_softmax (functional and SparseMatrix method)
    def soft(s):
        m = torch.stack([s[ei[0] == i].max() if (ei[0] == i).any() else s.new_zeros(()) for i in range(n)]).detach()
        e = torch.exp(s - m[ei[0]])
        return e / (seg_sum(e, ei[0], n) + 1e-8)[ei[0]]
    check(lambda s: tfg.nn.segment_softmax(s, ei[0].int(), n), Ref(soft), [w64], "segment_softmax")
    check(lambda s, x: tfg.SparseMatrix(ei.int(), s, [n, n]).segment_softmax(axis=-1) @ x,
          Ref(lambda s, x: seg_sum(x[ei[1]] * soft(s).unsqueeze(1), ei[0], n)), [w64, x64], "softmax @ x")
    # generic aggregate_neighbors route (a user mapper): gathers + user maths + HIP reducer, all tracked
    mapper = lambda rx, nx, edge_weight=None: (nx - rx) * edge_weight.unsqueeze(1)      # noqa: E731
    check(lambda x, w: tfg.nn.aggregate_neighbors(x, ei.int(), w, mapper, tfg.nn.mean_reducer, tfg.nn.sum_updater),
          Ref(lambda x, w: x + seg_sum((x[ei[1]] - x[ei[0]]) * w.unsqueeze(1), ei[0], n) /
              torch.bincount(ei[0], minlength=n).clamp(min=1).double().unsqueeze(1)), [x64, w64], "generic route")
    # edge_weight gradient through GraphSAGE's neighbour reduce
    ws = torch.tensor(rng.standard_normal((f, 4)), device=dev)
    wn = torch.tensor(rng.standard_normal((f, 4)), device=dev)
    check(lambda x, w: tfg.nn.sum_graph_sage(x, ei.int(), w, ws.float(), wn.float()),
          Ref(lambda x, w: torch.cat([x @ ws, seg_sum(x[ei[1]] * w.unsqueeze(1), ei[0], n) @ wn], 1)), [x64, w64],
          "sum_graph_sage d/dw")
    _ = proj


@pytest.mark.parametrize("m,ka,n", [(1, 1, 1), (37, 5, 3), (1000, 100, 256), (4099, 100, 40), (70001, 128, 128),
                                    (5000, 256, 256), (3000, 602, 64), (2708, 1433, 16), (999, 33, 300), (64, 32, 32),
                                    (700, 2500, 24), (3001, 7, 700), (9, 17, 16), (100003, 101, 41)])
@pytest.mark.parametrize("want_bias", [False, True])
def test_gemm_tn_weight_gradient_kernel(tfg, m, ka, n, want_bias):
    """tfgx_gemm_tn_f32 (dW = x^T g, db = column sums of g) vs float64, and the transpose kernel."""
    import torch
    from tf_geometric_amd.plan import gemm_tn, transpose
    gen = torch.Generator(device="cuda")
    gen.manual_seed(m * 31 + ka)
    x = torch.randn(m, ka, generator=gen, device="cuda")
    g = torch.randn(m, n, generator=gen, device="cuda")
    dW, db = gemm_tn(x, g, want_bias=want_bias)
    ref = (x.double().t() @ g.double())
    scale = (x.double().abs().t() @ g.double().abs())            # fp32 reduction over m terms: error ~ eps * sum |terms|
    assert float(((dW.double() - ref).abs() / (scale + 1e-30)).max()) < 3e-7
    if want_bias:
        rb = g.double().sum(0)
        assert float(((db.double() - rb).abs() / (g.double().abs().sum(0) + 1e-30)).max()) < 3e-7
    else:
        assert db is None
    dW2, _ = gemm_tn(x, g, want_bias=want_bias)
    assert torch.equal(dW, dW2)                                   # deterministic
    # gated form (the ReLU mask of the producing layer applied in registers) == the product with the masked gradient
    gate = torch.randn(m, n, generator=gen, device="cuda")
    gm = torch.where(gate > 0, g, torch.zeros_like(g))
    dWg, dbg = gemm_tn(x, g, want_bias=want_bias, gate=gate)
    dWm, dbm = gemm_tn(x, gm, want_bias=want_bias)
    assert torch.equal(dWg, dWm) and (dbg is None or torch.equal(dbg, dbm))
    assert torch.equal(transpose(x), x.t().contiguous())
    # strided views (a column block of a wider matrix) are honoured
    wide = torch.randn(m, n + 8, generator=gen, device="cuda")
    dW3, _ = gemm_tn(x, wide[:, 4:4 + n], want_bias=False)
    ref3 = x.double().t() @ wide[:, 4:4 + n].double()
    assert float(((dW3.double() - ref3).abs() / ((x.double().abs().t() @ wide[:, 4:4 + n].double().abs()) + 1e-30)).max()) < 3e-7


@pytest.mark.parametrize("f,weighted", [(8, True), (100, False), (128, True), (260, True)])
def test_max_gradient_push_equals_pull_and_autograd(tfg, oracle, f, weighted):
    """Mask form (default: per-edge winner bit masks, one gather per edge, deterministic) and push form of the segment-max gradient (arg positions saved by the training forward, N*F float atomics, rows with
    tied maxima walked exactly) vs the bit-reproducible pull kernel and vs float64 autograd (amax: ties share evenly, the
    TF rule).  The graph has duplicate edges and ReLU-style zero plateaus, i.e. plenty of ties, and an empty row."""
    from tf_geometric_amd import autograd as AG
    rng = np.random.Generator(np.random.PCG64(f))
    n = 500
    ei = oracle.synthetic_edges(n, 6000, seed=f)
    ei = ei[:, ei[0] != 7]
    ei = np.concatenate([ei, ei[:, :400]], axis=1)                     # duplicate edges: exact ties
    x = np.maximum(rng.standard_normal((n, f)), 0).astype(np.float32)  # many exact zeros
    w = (rng.integers(1, 3, ei.shape[1]) * 0.5).astype(np.float32) if weighted else None
    gout = rng.standard_normal((n, f)).astype(np.float32)
    mapper = tfg.nn.gcn_mapper if weighted else tfg.nn.identity_mapper

    def run(mode):
        AG.MAX_GRADIENT_MODE = mode
        try:
            xt = torch.tensor(x, device="cuda", requires_grad=True)
            out = tfg.nn.aggregate_neighbors(xt, ei, w, mapper, tfg.nn.max_reducer, tfg.nn.identity_updater)
            out.backward(torch.tensor(gout, device="cuda"))
            return out.detach().cpu().numpy(), xt.grad.cpu().numpy()
        finally:
            AG.MAX_GRADIENT_MODE = "mask"

    out_push, gx_push = run("push")
    out_pull, gx_pull = run("pull")
    out_mask, gx_mask = run("mask")
    assert np.array_equal(out_push, out_pull) and np.array_equal(out_mask, out_pull)
    assert_parity(gx_push, gx_pull, tol=2e-6, what="push vs pull max gradient")
    assert_parity(gx_mask, gx_pull, tol=2e-6, what="mask vs pull max gradient")
    assert np.array_equal(gx_mask, run("mask")[1])                     # the mask form is bit-reproducible
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wr = None if w is None else torch.tensor(w, dtype=torch.float64)
    ref = _ref_aggregate(xr, ei, wr, "max", n)
    ref.backward(torch.tensor(gout, dtype=torch.float64))
    assert_parity(gx_push, xr.grad.numpy(), tol=2e-5, what="push max gradient vs autograd")
    assert np.abs(gx_push[7]).max() >= 0 and np.array_equal(out_push[7], np.full(f, -3.4028234663852886e38, np.float32))


@pytest.mark.parametrize("threshold", [None, 64])
def test_hub_rows_in_the_backward_passes(tfg, oracle, threshold):
    """Power-law graphs: a destination with 6000 in-edges and a source with 5000 out-edges.  The GAT backward (dQ over hub
    destinations, dK / dV over hub sources) and the max-aggregation gradient (tie count over hub destinations, pull pass
    over hub sources) walk such rows chunk-wise and add the chunk partials in order — gradients equal float64 autograd,
    with the plan's own hub policy and with a forced low threshold that turns many rows into hubs."""
    import tf_geometric_amd.plan as P
    from tf_geometric_amd.plan import CsrPlan
    n, f = 1500, 12
    rng = np.random.Generator(np.random.PCG64(97))
    x = rng.standard_normal((n, f)).astype(np.float32)
    hub_dst = np.stack([np.full(6000, 7, np.int32), rng.integers(0, n, 6000, dtype=np.int32)])
    hub_src = np.stack([rng.integers(0, n, 5000, dtype=np.int32), np.full(5000, 11, np.int32)])
    ei = np.concatenate([hub_dst, oracle.synthetic_edges(n, 8000, seed=6), hub_src], axis=1).astype(np.int32)
    old = (P.HUB_THRESHOLD, P.HUB_CHUNK)
    P.HUB_THRESHOLD, P.HUB_CHUNK = threshold, (None if threshold is None else 48)
    try:
        plan = CsrPlan.build(ei, n, n)
        assert plan.hub_info() is not None and plan.transposed().hub_info() is not None
        cache = {"tfgx_csr_plan": plan}
        # GAT layer gradients
        heads, att, units = 4, 8, 16
        layer = tfg.layers.GAT(units, attention_units=att, num_heads=heads, activation=tfg.relu)
        layer._maybe_build([x])
        ws = {"query_kernel": oracle.glorot_uniform(rng, f, att), "query_bias": (rng.standard_normal(att) * 0.2).astype(np.float32),
              "key_kernel": oracle.glorot_uniform(rng, f, att), "key_bias": (rng.standard_normal(att) * 0.2).astype(np.float32),
              "kernel": oracle.glorot_uniform(rng, f, units), "bias": (rng.standard_normal(units) * 0.1).astype(np.float32)}
        layer.set_weights(**ws)
        layer.trainable(True)
        xt = torch.tensor(x, device="cuda", requires_grad=True)
        out = layer([xt, ei], cache=cache)
        gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
        out.backward(gout)
        r = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
        xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        ref = _ref_gat(xr, ei, r["query_kernel"], r["query_bias"], r["key_kernel"], r["key_bias"], r["kernel"], r["bias"], heads, n)
        ref.backward(gout.double().cpu())
        assert_parity(out.detach().cpu().numpy(), ref.detach().numpy(), tol=2e-5, what="hub gat forward")
        assert_parity(xt.grad.cpu().numpy(), xr.grad.numpy(), tol=2e-4, what="hub gat d/dx")
        for k in ws:
            assert_parity(getattr(layer, k).grad.cpu().numpy(), r[k].grad.numpy(), tol=5e-4, what="hub gat d/d" + k)
        # max aggregation gradient (ties included: ReLU-style zero plateaus)
        xm = np.maximum(rng.standard_normal((n, f)), 0).astype(np.float32)
        xt2 = torch.tensor(xm, device="cuda", requires_grad=True)
        o2 = tfg.nn.aggregate_neighbors(xt2, ei, None, tfg.nn.identity_mapper, tfg.nn.max_reducer, tfg.nn.identity_updater)
        g2 = torch.tensor(rng.standard_normal((n, f)).astype(np.float32), device="cuda")
        o2.backward(g2)
        xr2 = torch.tensor(xm, dtype=torch.float64, requires_grad=True)
        r2 = _ref_aggregate(xr2, ei, None, "max", n)
        r2.backward(g2.double().cpu())
        assert np.array_equal(o2.detach().cpu().numpy(), r2.detach().numpy().astype(np.float32))
        assert_parity(xt2.grad.cpu().numpy(), xr2.grad.numpy(), tol=2e-5, what="hub max d/dx")
    finally:
        P.HUB_THRESHOLD, P.HUB_CHUNK = old


@pytest.mark.parametrize("op", ["sum", "mean", "max"])
@pytest.mark.parametrize("weighted", [True, False])
def test_hip_gradients_match_tf_registered_gradients(tfg, op, weighted):
    """The HIP backward vs oracle/tf_gradients.py — TensorFlow's REGISTERED gradients (math_grad.py:
    _UnsortedSegmentSumGrad, _UnsortedSegmentMinOrMaxGrad, _GatherV2Grad ...) chained along the reference's forward lines
    (nn/kernel/map_reduce.py:45-73), pinned by hand-derived known answers in tests/test_tf_gradient_kats.py.  Graph with
    duplicated edges and quantised features (many tied maxima) and empty rows."""
    from oracle import tf_gradients as G
    rng = np.random.Generator(np.random.PCG64(31))
    n, e, f = 400, 5000, 24
    ei = rng.integers(0, n - 20, size=(2, e)).astype(np.int32)
    ei = np.concatenate([ei, ei[:, :1500]], axis=1)
    x = (np.round(rng.standard_normal((n, f)) * 2) / 2).astype(np.float32)
    w = (rng.integers(1, 3, ei.shape[1]) * 0.5).astype(np.float32) if weighted else None
    g = rng.standard_normal((n, f)).astype(np.float32)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    wt = torch.tensor(w, device="cuda", requires_grad=True) if weighted else None
    mapper = tfg.nn.gcn_mapper if weighted else tfg.nn.identity_mapper
    out = tfg.nn.aggregate_neighbors(xt, ei, wt, mapper, getattr(tfg.nn, op + "_reducer"), tfg.nn.sum_updater)
    out.backward(torch.tensor(g, device="cuda"))
    ref_out, dx, dw = G.aggregate_neighbors_grad(x, ei, w, op, "sum", g)
    assert_parity(out.detach().cpu().numpy(), ref_out, what="forward " + op)
    assert_parity(xt.grad.cpu().numpy(), dx, tol=2e-5, what="d/dx vs TF registered gradients, " + op)
    if weighted:
        assert_parity(wt.grad.cpu().numpy(), dw, tol=2e-5, what="d/dw vs TF registered gradients, " + op)


def test_hip_segment_softmax_gradient_matches_tf_registered_gradients(tfg):
    """nn/kernel/segment.py:26-33 under tf.GradientTape: max under stop_gradient, +1e-8 in the denominator."""
    from oracle import tf_gradients as G
    rng = np.random.Generator(np.random.PCG64(32))
    n, e = 200, 3000
    ids = rng.integers(0, n - 10, size=e).astype(np.int32)
    for shape in ((e,), (e, 4)):
        s = (rng.standard_normal(shape) * 3).astype(np.float32)
        g = rng.standard_normal(shape).astype(np.float32)
        st = torch.tensor(s, device="cuda", requires_grad=True)
        score = tfg.nn.segment_softmax(st, ids, n)
        score.backward(torch.tensor(g, device="cuda"))
        ref_score, ds = G.segment_softmax_grad(s, ids, n, g)
        assert_parity(score.detach().cpu().numpy(), ref_score, what="segment_softmax")
        assert_parity(st.grad.cpu().numpy(), ds, tol=2e-5, what="d segment_softmax / d data vs TF registered gradients")


def test_hip_gcn_layer_gradients_match_tf_registered_gradients(tfg, oracle):
    from oracle import tf_gradients as G
    x, ei, w, rng = _graph(oracle, seed=9)
    n, f = x.shape
    units = 12
    k = oracle.glorot_uniform(rng, f, units)
    b = (rng.standard_normal(units) * 0.1).astype(np.float32)
    layer = tfg.layers.GCN(units, activation=tfg.relu)
    layer._maybe_build([x])
    layer.set_weights(kernel=k, bias=b)
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    out = layer([xt, ei, w], cache={})
    g = rng.standard_normal((n, units)).astype(np.float32)
    out.backward(torch.tensor(g, device="cuda"))
    idx, nw = oracle.gcn_norm_adj(ei, w, n)
    ref_out, dx, dk, db = G.gcn_layer_grad(x, idx, nw, k, b, True, g)
    assert_parity(out.detach().cpu().numpy(), ref_out, what="gcn forward")
    assert_parity(xt.grad.cpu().numpy(), dx, tol=2e-5, what="gcn d/dx")
    assert_parity(layer.kernel.grad.cpu().numpy(), dk, tol=2e-5, what="gcn d/dkernel")
    assert_parity(layer.bias.grad.cpu().numpy(), db, tol=2e-5, what="gcn d/dbias")


@pytest.mark.parametrize("f,weighted", [(100, True), (64, False), (32, True), (256, False), (36, True), (512, False), (384, True)])
def test_tracked_max_forward_equals_the_arg_kernel(tfg, oracle, f, weighted):
    """tfgx_reduce_args.track (the tuned segment-reduce walk with the tie count and the first maximal edge's position
    tracked online, packed count << 16 | row-relative position) vs tfgx_segment_max_with_arg_f32: identical maxima,
    counts and positions — duplicated edges and quantised features (many ties), an empty row; and the mask-form gradient
    computed from the packed array is bit-identical to the one computed from the two arrays."""
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd import autograd as AG
    from tf_geometric_amd.plan import CsrPlan, segment_reduce, can_track
    rng = np.random.Generator(np.random.PCG64(f))
    n = 900
    ei = oracle.synthetic_edges(n, 12000 if f <= 256 else 40000, seed=f)      # rows wider than 256 columns track through the
    ei = ei[:, ei[0] != 9]                                                     # column-block walk, which dense plans take
    ei = np.concatenate([ei, ei[:, :4000]], axis=1)
    x = np.round(rng.standard_normal((n, f)).astype(np.float32) * 2) / 2
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    xd = L.as_f32(x)
    w = (torch.randint(1, 3, (plan.num_edges,), device="cuda").float() * 0.5) if weighted else None
    assert can_track(plan, xd, f)
    lib = L.require_gpu()
    out0, cnt0 = torch.empty((n, f), device="cuda"), torch.empty((n, f), device="cuda")
    arg0 = torch.empty((n, f), dtype=torch.int32, device="cuda")
    L.check(lib.tfgx_segment_max_with_arg_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w), n, L.ptr(xd), f, f,
                                              L.ptr(out0), f, L.ptr(cnt0), f, L.ptr(arg0), f, L.stream_ptr()), "with_arg")
    out1 = torch.empty((n, f), device="cuda")
    packed = torch.empty((n, f), dtype=torch.int32, device="cuda")
    segment_reduce(plan, xd, L.MAX, w_csr=w, out=out1, track=packed)
    pk = packed.to(torch.int64) & 0xFFFFFFFF
    cnt1 = (pk >> 16).float()
    pos1 = torch.where(cnt1 > 0, plan.row_ptr[:-1].long().unsqueeze(1) + (pk & 0xFFFF), torch.full_like(pk, -1))
    assert torch.equal(out0, out1) and torch.equal(cnt0, cnt1) and torch.equal(arg0.long(), pos1)
    assert float(cnt1.max()) >= 2 and float(cnt1[9].abs().max()) == 0 and int((pk[9] & 0xFFFF).min()) == 0xFFFF
    # gradients: the autograd path now takes the packed form; it must equal the two-array mask form bit for bit
    g = torch.randn(n, f, device="cuda")
    xt = xd.clone().requires_grad_(True)
    AG.aggregate(plan, xt, L.MAX, w_csr=w).backward(g)
    pt, t2d = AG._transposed(plan)
    w_t = AG._transposed_weights(plan, w, t2d)
    ws_bytes = lib.tfgx_segment_max_backward_mask_workspace_bytes(n, plan.num_edges, f)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    gx = torch.empty((n, f), device="cuda")
    L.check(lib.tfgx_segment_max_backward_mask_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w), n, plan.num_edges,
                                                   L.ptr(xd), f, f, L.ptr(out0), f, L.ptr(g), f, L.ptr(cnt0), f,
                                                   L.ptr(arg0), f, L.ptr(pt.row_ptr), L.ptr(pt.col), L.ptr(w_t), L.ptr(t2d),
                                                   n, L.ptr(gx), f, L.ptr(ws), ws_bytes, L.stream_ptr()), "mask (two arrays)")
    assert torch.equal(xt.grad, gx)


@pytest.mark.parametrize("f,weighted,k1", [(100, True, 3), (64, False, 2), (32, True, 4), (256, False, 3)])
def test_tracked_max_span_by_span_equals_one_launch(tfg, oracle, f, weighted, k1):
    """tfgx_reduce_args.track with accumulate = 1: a row reduced in k1 launches over consecutive sub-spans (what the sharded
    path does: own-source edges under the halo exchange, then the sub-span of each round) ends with the SAME maxima and the
    same packed (tie count, first position) as one launch over the whole row — ties inside and ACROSS sub-spans (quantised
    features, duplicated edges), empty sub-spans, an empty row; and max_passes on the autograd function reproduces the
    single-launch gradient bit for bit."""
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd import autograd as AG
    from tf_geometric_amd.plan import CsrPlan, segment_reduce
    rng = np.random.Generator(np.random.PCG64(1000 + f))
    n = 700
    ei = oracle.synthetic_edges(n, 9000, seed=f + 1)
    ei = ei[:, ei[0] != 5]
    ei = np.concatenate([ei, ei[:, :3000]], axis=1)
    x = np.round(rng.standard_normal((n, f)).astype(np.float32) * 2) / 2
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    xd = L.as_f32(x)
    w = (torch.randint(1, 3, (plan.num_edges,), device="cuda").float() * 0.5) if weighted else None
    rp = plan.row_ptr.long()
    deg = rp[1:] - rp[:-1]
    cuts = torch.sort(torch.rand(n, k1 - 1, device="cuda"), dim=1).values        # random cut fractions per row
    cuts[::7] = 0.0                                                             # some rows: empty leading sub-spans
    cuts[3::11] = 1.0                                                           # ... or empty trailing ones
    inner = rp[:-1].unsqueeze(1) + torch.floor(cuts * deg.unsqueeze(1).float()).long()
    rpk = torch.cat([torch.cat([rp[:-1].unsqueeze(1), inner], 1).reshape(-1), rp[-1:]]).to(torch.int32).contiguous()
    out1 = torch.empty((n, f), device="cuda")
    pk1 = torch.empty((n, f), dtype=torch.int32, device="cuda")
    segment_reduce(plan, xd, L.MAX, w_csr=w, out=out1, track=pk1)

    def passes(x2, w2, out, packed):
        for k in range(k1):
            segment_reduce(plan, x2, L.MAX, w_csr=w2, out=out, track=packed, accumulate=k > 0, row_begin=rpk[k:],
                           row_end=rpk[k + 1:], rp_stride=k1, col=plan.col, n_dst=n, track_row_begin=rpk)

    out2 = torch.empty((n, f), device="cuda")
    pk2 = torch.full((n, f), 12345, dtype=torch.int32, device="cuda")
    passes(xd, w, out2, pk2)
    assert torch.equal(out1, out2)
    assert torch.equal(pk1, pk2), int((pk1 != pk2).sum())
    assert int(((pk1.long() & 0xFFFFFFFF) >> 16).max()) >= 2
    g = torch.randn(n, f, device="cuda")
    xa, xb = xd.clone().requires_grad_(True), xd.clone().requires_grad_(True)
    AG.aggregate(plan, xa, L.MAX, w_csr=w).backward(g)
    AG.aggregate(plan, xb, L.MAX, w_csr=w, max_passes=passes).backward(g)
    assert torch.equal(xa.grad, xb.grad)
    # halo-first backward (tfgx_segment_max_backward_mask_phases_f32): masks built once, applied to the source rows
    # [n_first, n) first — the hook sees their final gradients (and, in the sharded path, sends them on their way) — then
    # to [0, n_first): the same gradient bit for bit
    seen = {}

    def hook(j, gx):
        seen[j] = gx[n // 3 + j * 100:(n if j == 1 else n // 3 + 100)].clone()
    passes.halo_first = (n // 3, [(n // 3, n // 3 + 100), (n // 3 + 100, n)], hook)     # two windows, as two exchange rounds
    xc = xd.clone().requires_grad_(True)
    AG.aggregate(plan, xc, L.MAX, w_csr=w, max_passes=passes).backward(g)
    assert torch.equal(xa.grad, xc.grad)
    assert torch.equal(seen[0], xa.grad[n // 3:n // 3 + 100]) and torch.equal(seen[1], xa.grad[n // 3 + 100:])


@pytest.mark.parametrize("kind", ["gcn", "MeanGraphSage"])
def test_training_on_the_promoted_static_layout_gives_the_same_bits(tfg, oracle, kind):
    """A layer's second call with the same unchanged feature tensor promotes it to the static layout (plan.static_rows);
    the TRAINING forward then runs the fused launch on split rows + the per-edge tail stream.  Same FMA chains, same
    projection order: output and every gradient are bit-identical to the run on the plain table.  (The promotion is
    restricted to matrices far larger than the caches; the size test is lifted here to exercise the route at test size.)"""
    from tf_geometric_amd import plan as P
    x, ei, w, rng = _graph(oracle, n=3000, e=40000, f=100, seed=41)
    n, units = x.shape[0], 256
    xt = tfg._lib.as_f32(x)
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    layer = tfg.layers.GCN(units, activation=tfg.relu) if kind == "gcn" else getattr(tfg.layers, kind)(units, activation=tfg.relu)
    layer._maybe_build([x])
    layer.trainable(True)

    def run(cache):
        for p_ in layer.parameters():
            p_.grad = None
        out = layer([xt, ei, w], cache=cache)
        out.backward(gout)
        return out.detach().clone(), [p_.grad.clone() for p_ in layer.parameters()]

    wanted = P.SplitRows.wanted
    P.SplitRows.wanted = staticmethod(lambda n_, F: F % 4 == 0 and 32 < F <= 128 and F % 32 != 0)
    try:
        cache = {}
        o1, g1 = run(cache)                                  # first sighting: the plain table
        assert cache.get("tfgx_static_rows") is None
        before = dict(P.FUSED_STATS)
        o2, g2 = run(cache)                                  # second call: promoted, fused launch on the split rows
        assert cache["tfgx_static_rows"][1] is not None and cache["tfgx_static_rows"][1].edge_tail is not None
        assert P.FUSED_STATS["launches"] == before["launches"] + 1 and P.FUSED_STATS["with_side_output"] == before["with_side_output"] + 1
    finally:
        P.SplitRows.wanted = wanted
    assert torch.equal(o1, o2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("m,n", [(1, 1), (5, 47), (70001, 47), (300000, 41), (4099, 256), (100003, 7), (2500, 1030), (0, 9)])
def test_column_sums_kernel(tfg, m, n):
    """tfgx_column_sum_f32 (the bias gradient where the bias rode in an aggregation epilogue) vs float64; strided input;
    deterministic.  Odd widths are the point: torch's g.sum(0) runs a [2.4 M, 47] gradient at 24 GB/s."""
    from tf_geometric_amd.plan import column_sums
    g = torch.Generator(device="cuda")
    g.manual_seed(m + n)
    wide = torch.randn(m, n + 3, generator=g, device="cuda")
    x = wide[:, 1:n + 1]                                      # row stride n + 3
    got = column_sums(x)
    assert got.shape == (n,) and torch.equal(got, column_sums(x))
    ref = x.double().sum(0)
    scale = max(1.0, float(m) ** 0.5)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, float(x.abs().max()) if m else 1.0)


@pytest.mark.parametrize("f_in,units,n,e,long_rows", [(100, 256, 3000, 150000, False), (4, 64, 500, 6000, True),
                                                      (124, 128, 800, 30000, True), (36, 256, 700, 9000, False)])
def test_pool_mlp_max_weight_gradient_from_destination_rows(tfg, oracle, f_in, units, n, e, long_rows):
    """MaxPoolGraphSage, layer-0 form (x carries no gradient): the MLP + max operator whose backward goes from the destination
    rows straight to dW / db (tfgx_pool_mlp_max_wgrad_f32) against (i) the composed form (winner masks -> dh -> ReLU mask ->
    x^T dh) and (ii) float64 autograd with TensorFlow's tie rule.  Quantised features and duplicated edges (tied maxima between
    DIFFERENT sources and between copies of one edge), rows without in-edges, rows longer than the 96-edge LDS chunk."""
    from tf_geometric_amd import _lib as L, autograd as AG
    rng = np.random.Generator(np.random.PCG64(f_in + units))
    ei = oracle.synthetic_edges(n, e, seed=f_in)
    ei = ei[:, ei[0] != 7]                                              # row 7: no in-edges
    ei = np.concatenate([ei, ei[:, :e // 8]], axis=1)                   # duplicated edges
    if long_rows:
        extra = np.stack([np.full(700, 5, np.int32), rng.integers(0, n, 700).astype(np.int32)])
        ei = np.concatenate([ei, extra, np.stack([np.full(97, 11, np.int32), rng.integers(0, n, 97).astype(np.int32)])], axis=1)
    x = (np.round(rng.standard_normal((n, f_in)) * 2) / 2).astype(np.float32)          # quantised: exact ties between sources
    w1 = np.ones(ei.shape[1], np.float32)
    ku = units // 2
    ws = dict(self_kernel=oracle.glorot_uniform(rng, f_in, ku),
              mlp_kernel=(np.round(oracle.glorot_uniform(rng, f_in, 4 * ku) * 8) / 8).astype(np.float32),
              mlp_bias=(np.round(rng.standard_normal(4 * ku)) * 0.25).astype(np.float32),
              neighs_kernel=oracle.glorot_uniform(rng, 4 * ku, ku), bias=(rng.standard_normal(units) * 0.1).astype(np.float32))
    gout = torch.tensor(rng.standard_normal((n, units)).astype(np.float32), device="cuda")
    gout[7] = 0.0                        # the row without in-edges pools float32 lowest: no gradient through its overflow

    def run(fused):
        AG.POOL_MLP_MAX_FUSED = fused
        try:
            layer = tfg.layers.MaxPoolGraphSage(units, activation=tfg.relu, concat=True)
            layer._maybe_build([x])
            layer.set_weights(**ws)
            layer.trainable(True)
            out = layer([L.as_f32(x), ei, w1], cache={})
            out.backward(gout)
            return out.detach(), {k: v.grad.clone() for k, v in layer.weights.items()}
        finally:
            AG.POOL_MLP_MAX_FUSED = True
    lib = L.require_gpu()
    assert lib.tfgx_pool_mlp_max_wgrad_applies(f_in, 4 * ku)
    out_f, g_f = run(True)
    out_c, g_c = run(False)
    assert torch.equal(out_f, out_c)
    for k in ws:
        scale = float(g_c[k].abs().max()) + 1.0
        assert float((g_f[k] - g_c[k]).abs().max()) <= 2e-5 * scale, k
    # float64 autograd, ties shared evenly (math_grad._UnsortedSegmentMinOrMaxGrad)
    row, col = torch.from_numpy(ei[0]).long(), torch.from_numpy(ei[1]).long()
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in ws.items()}
    xr = torch.tensor(x, dtype=torch.float64)
    h = torch.relu(xr @ t["mlp_kernel"] + t["mlp_bias"])
    red = torch.full((n, 4 * ku), -3.4028234663852886e38, dtype=torch.float64).scatter_reduce(
        0, row[:, None].expand(-1, 4 * ku), h[col], "amax", include_self=True)
    ref = torch.relu(torch.cat([xr @ t["self_kernel"], red @ t["neighs_kernel"]], dim=1) + t["bias"])
    keep = torch.ones(n, dtype=torch.bool)
    keep[7] = False
    ref.backward(gout.double().cpu())
    assert_parity(out_f[keep].cpu().numpy(), ref.detach()[keep].numpy(), what="max-pool layer forward")
    for k in ("mlp_kernel", "mlp_bias"):
        scale = float(t[k].grad.abs().max()) + 1.0
        assert float((g_f[k].double().cpu() - t[k].grad).abs().max()) <= 2e-5 * scale, k
