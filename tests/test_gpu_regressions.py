# coding=utf-8
"""Regression tests for the round-2 advisor findings: each one reproduces the reported misbehaviour."""
import numpy as np
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _graph(oracle, n, e, f, seed):
    ei = oracle.synthetic_edges(n, e, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    return rng.standard_normal((n, f)).astype(np.float32), ei, rng


def _softmax_np(h):
    h = h.astype(np.float64)
    ex = np.exp(h - h.max(1, keepdims=True))
    return (ex / ex.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("units", [41, 7, 43])
@pytest.mark.parametrize("training", [False, True])
def test_single_head_gat_row_wise_activation_sees_only_real_columns(tfg, oracle, units, training):
    """GAT(41, num_heads=1) runs zero-padded to a multiple of four value columns (nn/conv/gat.py).  A ROW-WISE activation
    (softmax over the class scores — the natural choice for the output layer this path targets) must run on the unpadded
    result, as the reference applies it last (nn/conv/gat.py:119-120): each padded column would add exp(0) to the
    denominator."""
    x, ei, rng = _graph(oracle, 300, 3000, 24, seed=units)
    wq, wk, wv = oracle.glorot_uniform(rng, 24, 1), oracle.glorot_uniform(rng, 24, 1), oracle.glorot_uniform(rng, 24, units)
    bq, bk = np.float32([0.1]), np.float32([-0.05])
    b = (rng.standard_normal(units) * 0.1).astype(np.float32)
    act = lambda h: torch.softmax(h, dim=-1)       # noqa: E731
    layer = tfg.layers.GAT(units, attention_units=1, num_heads=1, activation=act)
    layer._maybe_build([x])
    layer.set_weights(query_kernel=wq, query_bias=bq, key_kernel=wk, key_bias=bk, kernel=wv, bias=b)
    if training:
        layer.trainable(True)
    got = layer([x, ei])
    assert tuple(got.shape) == (300, units)
    ref = oracle.gat(x, ei, wq, bq, "relu", wk, bk, "relu", wv, b, _softmax_np, num_heads=1)
    assert_parity(got.detach().cpu().numpy(), ref, what="GAT({}) + softmax activation".format(units))
    assert_parity(got.detach().sum(1).cpu().numpy(), np.ones(300), what="rows sum to one")
    if training:
        got.square().sum().backward()
        assert layer.kernel.grad is not None and tuple(layer.kernel.grad.shape) == (24, units)


@pytest.mark.parametrize("f_out", [7, 47, 20])
def test_k_hop_convolutions_return_dense_tensors(tfg, oracle, f_out):
    """The line-friendly row stride (F = 7 -> 8, 47 -> 48) is for the links INSIDE a k-hop chain; what the public
    functions return is a dense [n, F] tensor (.view(-1) works, ld == F for DLPack / C-ABI consumers)."""
    x, ei, rng = _graph(oracle, 200, 1500, 12, seed=f_out)
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    k9 = oracle.glorot_uniform(rng, 12, f_out)
    outs = {
        "sgc": tfg.nn.sgc(x, ei, w, 2, k9),
        "appnp": tfg.nn.appnp(x, ei, w, [k9], [np.zeros(f_out, np.float32)], k=3),
        "ssgc": tfg.nn.ssgc(x, ei, w, [k9], [np.zeros(f_out, np.float32)], k=3),
        "tagcn": tfg.nn.tagcn(x, ei, w, 2, oracle.glorot_uniform(rng, 36, f_out)),
        "chebynet": tfg.nn.chebynet(x, ei, w, 3, [oracle.glorot_uniform(rng, 12, f_out) for _ in range(3)]),
        "ssgc-plain": tfg.nn.ssgc(rng.standard_normal((200, f_out)).astype(np.float32), ei, None, None, None, k=2),
    }
    for name, t in outs.items():
        assert t.is_contiguous() and t.stride(0) == t.shape[1], name
        t.view(-1)


def test_set_weights_copies_the_callers_tensor(tfg, oracle):
    """layer.set_weights(kernel=<float32 device tensor>) must not alias it: requires_grad_ / optimizer steps would write
    into user data."""
    x, ei, rng = _graph(oracle, 100, 600, 8, seed=3)
    mine = torch.tensor(oracle.glorot_uniform(rng, 8, 5), device="cuda")
    keep = mine.clone()
    layer = tfg.layers.GCN(5)
    layer._maybe_build([x])
    layer.trainable(True)
    layer.set_weights(kernel=mine)
    assert layer.kernel.data_ptr() != mine.data_ptr() and not mine.requires_grad
    opt = torch.optim.SGD(layer.parameters(), lr=0.5)
    layer([x, ei], cache={}).square().sum().backward()
    opt.step()
    assert torch.equal(mine, keep) and not torch.equal(layer.kernel.detach(), keep)
    # by attribute name too (MaxPoolGraphSage: variable "mlp_kernel" lives in attribute neighbor_mlp_kernel)
    sage = tfg.layers.MaxPoolGraphSage(6)
    sage._maybe_build([x])
    new = torch.randn_like(sage.weights["mlp_kernel"])
    sage.set_weights(mlp_kernel=new)
    assert torch.equal(sage.neighbor_mlp_kernel, new) and sage.neighbor_mlp_kernel.data_ptr() != new.data_ptr()
    assert sage.weights["mlp_kernel"] is sage.neighbor_mlp_kernel


def test_plan_metadata_is_not_built_inside_a_capture(tfg, oracle):
    """plan.row_order() / hub_info() synchronise on first use; when the first use happens under hipGraph capture they
    return None (the launch runs in natural order / inline) instead of breaking the capture."""
    from tf_geometric_amd import synthetic
    from tf_geometric_amd.plan import CsrPlan, segment_reduce
    L = tfg._lib
    n = 1 << 14
    ei = synthetic.rmat_edges(n, 400000, 3, torch.device("cuda"))
    x = torch.randn(n, 16, device="cuda")
    eager = segment_reduce(CsrPlan.build(ei, n, n), x, L.SUM)
    plan = CsrPlan.build(ei, n, n)                      # fresh plan: nothing lazy computed yet
    out = torch.empty_like(eager)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        segment_reduce(plan, x, L.SUM, out=out)
    graph.replay()
    torch.cuda.synchronize()
    # inline walk of the hub rows vs the eager plan's chunked walk: the same 10^3..10^4-term fp32 sums in another order
    assert torch.allclose(out, eager, rtol=1e-4, atol=2e-3), "captured first use of a skewed plan"
    assert plan.row_order() is not None and plan.hub_info() is not None       # computed by the next eager call


def test_dynamic_lambda_max_converges_beyond_the_krylov_window(tfg, oracle):
    """laplacian_max_eigenvalue: n far above the 96-step Arnoldi window; the Ritz residual is checked (restart when it is
    not small) and the value agrees with a dense eigen-solve of the same Laplacian (the reference: ARPACK,
    utils/graph_utils.py:884-909)."""
    from tf_geometric_amd.nn.conv.propagation import chebynet_norm_edge, laplacian_max_eigenvalue
    n = 1500
    rng = np.random.Generator(np.random.PCG64(11))
    a, b = rng.integers(0, n, 6000), rng.integers(0, n, 6000)
    keep = a != b
    lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    _, first = np.unique(lo * n + hi, return_index=True)
    lo, hi = lo[first], hi[first]
    ei = np.stack([np.concatenate([lo, hi]), np.concatenate([hi, lo])]).astype(np.int32)
    wu = rng.uniform(0.5, 1.5, lo.size).astype(np.float32)
    w = np.concatenate([wu, wu])
    for nt in ("sym", "rw", None):
        got = laplacian_max_eigenvalue(chebynet_norm_edge(ei, n, w, nt), nt)
        ref = oracle.laplacian_max_eigenvalue(ei, n, w, nt)
        info = laplacian_max_eigenvalue.last
        assert info["rel_residual"] <= 1e-5 or info["steps"] >= n, info
        assert abs(got - ref) <= 1e-4 * abs(ref), (nt, got, ref, info)
    # a tiny window forces restarts and still converges
    got = laplacian_max_eigenvalue(chebynet_norm_edge(ei, n, w, "sym"), "sym", steps=12, restarts=40)
    assert laplacian_max_eigenvalue.last["restarts"] >= 1
    assert abs(got - oracle.laplacian_max_eigenvalue(ei, n, w, "sym")) <= 1e-4 * abs(got)


def test_gcn_layer_validates_num_splits_at_build_time(tfg, oracle):
    """layers/conv/gcn.py:21-23 of the reference: build() computes the split and raises for an impossible num_splits
    (utils/tf_sparse_utils.py:71-90); a valid one does not change the output."""
    x, ei, rng = _graph(oracle, 120, 900, 10, seed=4)
    bad = tfg.layers.GCN(9, num_splits=4)                  # 9 columns into 4 parts: ceil = 3 -> only 3 parts
    with pytest.raises(Exception, match="cannot split H"):
        bad([x, ei])
    with pytest.raises(Exception, match="cannot provide both"):
        tfg.layers.GCN(9, num_splits=3, num_or_size_splits=[3, 3, 3])
    k = oracle.glorot_uniform(rng, 10, 9)
    outs = []
    for kw in (dict(), dict(num_splits=3), dict(num_splits=2), dict(num_or_size_splits=[4, 5])):
        layer = tfg.layers.GCN(9, use_bias=False, **kw)
        layer._maybe_build([x])
        layer.set_weights(kernel=k)
        outs.append(layer([x, ei]))
    assert tfg.layers.GCN(9, num_splits=3)._maybe_build([x]) is None
    lay = tfg.layers.GCN(9, num_splits=2)
    lay._maybe_build([x])
    assert lay.num_or_size_splits == [5, 4]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    nk = tfg.layers.GCN(9, use_kernel=False, num_splits=5)  # no kernel: the split applies to the F = 10 input columns
    nk._maybe_build([x])
    assert nk.num_or_size_splits == 5


def _two_layer(tfg, kind, units, classes):
    if kind == "gcn":
        return tfg.layers.GCN(units, activation=tfg.relu), tfg.layers.GCN(classes)
    if kind == "mean_sage":
        return tfg.layers.MeanGraphSage(units, activation=tfg.relu), tfg.layers.MeanGraphSage(classes, activation=None)
    if kind == "max_pool_sage":
        return tfg.layers.MaxPoolGraphSage(units, activation=tfg.relu), tfg.layers.MaxPoolGraphSage(classes, activation=None)
    return (tfg.layers.GAT(units, attention_units=8, num_heads=4, activation=tfg.relu),
            tfg.layers.GAT(classes, attention_units=4, num_heads=1))


@pytest.mark.parametrize("kind,opt_name", [("gcn", "adam"), ("gcn", "sgd"), ("mean_sage", "adam"), ("max_pool_sage", "adam"),
                                           ("gat", "adam")])
def test_captured_train_step_equals_the_eager_loop(tfg, oracle, kind, opt_name):
    """tfg.CapturedTrainStep: zero-grad + forward + loss + backward (the kernels' own backward) + optimizer update replayed
    from ONE hipGraph.  Step k of the replay must leave the weights where step k of the eager loop leaves them — including
    the constructor's warm-up steps being rolled back (parameters and optimizer state).  The role tf.function plays around
    the reference's training forward (demo/demo_gcn.py:64-83)."""
    n, e, f, units, classes = 3000, 24000, 32, 16, 8
    x_np, ei, rng = _graph(oracle, n, e, f, seed=11)
    x = tfg._lib.as_f32(x_np)
    w = np.ones(ei.shape[1], np.float32)
    labels = torch.as_tensor(rng.integers(0, classes, n), device=x.device)
    idx = torch.arange(0, n, 3, device=x.device)

    def make():
        cache = {}
        l0, l1 = _two_layer(tfg, kind, units, classes)
        inputs = (lambda h: [h, ei]) if kind in ("gcn", "gat") else (lambda h: [h, ei, w])
        with torch.no_grad():
            l1(inputs(l0(inputs(x), cache=cache)), cache=cache)
        l0.trainable(True)
        l1.trainable(True)
        return l0, l1, (lambda: torch.nn.functional.cross_entropy(l1(inputs(l0(inputs(x), cache=cache)), cache=cache)[idx],
                                                                 labels[idx]))

    a0, a1, loss_a = make()
    b0, b1, loss_b = make()
    for la, lb in ((a0, b0), (a1, b1)):
        lb.set_weights(**{k: v.detach() for k, v in la.weights.items() if v is not None})
        lb.trainable(True)

    def optimizer(layers):
        ps = layers[0].parameters() + layers[1].parameters()
        if opt_name == "adam":
            return torch.optim.Adam(ps, lr=1e-2, capturable=True)
        return torch.optim.SGD(ps, lr=5e-2, momentum=0.9)

    opt_a, opt_b = optimizer((a0, a1)), optimizer((b0, b1))
    losses_a = []
    for _ in range(6):
        opt_a.zero_grad(set_to_none=True)
        la = loss_a()
        la.backward()
        opt_a.step()
        losses_a.append(float(la.detach()))
    before = [p.detach().clone() for p in b0.parameters() + b1.parameters()]
    step = tfg.CapturedTrainStep(loss_b, opt_b)
    for p, q in zip(b0.parameters() + b1.parameters(), before):       # the warm-up left no trace
        assert torch.equal(p.detach(), q)
    losses_b = [float(step().detach()) for _ in range(6)]
    assert losses_a[-1] < losses_a[0]                                  # it trains
    np.testing.assert_allclose(losses_b, losses_a, rtol=2e-5, atol=1e-6)
    for p, q in zip(a0.parameters() + a1.parameters(), b0.parameters() + b1.parameters()):
        assert_parity(q.detach().cpu().numpy(), p.detach().cpu().numpy(), tol=2e-5, what="{} weights after 6 steps".format(kind))


def test_captured_train_step_needs_a_capturable_optimizer(tfg, oracle):
    x_np, ei, rng = _graph(oracle, 200, 1500, 8, seed=3)
    layer = tfg.layers.GCN(4)
    x = tfg._lib.as_f32(x_np)
    layer([x, ei])
    layer.trainable(True)
    with pytest.raises(ValueError, match="capturable"):
        tfg.CapturedTrainStep(lambda: layer([x, ei]).sum(), torch.optim.Adam(layer.parameters(), lr=1e-2))


def test_captured_gat_attention_dropout_draws_a_new_mask_on_every_replay(tfg, oracle):
    """a dropout seed passed BY VALUE is frozen into a hipGraph — every replay of a captured step then
    drops the same edges.  Under capture the seed lives on the device (nn/conv/gat.new_drop_seed: the captured sequence
    advances the per-device seed stream and snapshots it per layer call; kernels read tfgx_gat_args.drop_seed_dev), so
    (1) replays differ from one another, (2) each replay equals the EAGER layer run with that replay's seed value, and
    (3) a captured training step of a GAT with edge_drop_rate > 0 sees a different loss on replays with frozen weights."""
    from tf_geometric_amd.nn.conv import gat as G
    from tf_geometric_amd.plan import CsrPlan
    n, e, H = 2000, 30000, 2
    x_np, ei, rng = _graph(oracle, n, e, 16, seed=13)
    plan = CsrPlan.build(tfg._lib.as_i32(ei), n, n)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    Q, K, V = (torch.randn(n, 8, generator=g, device="cuda") for _ in range(3))
    G.new_drop_seed(Q.device)                                   # eager use creates the device seed stream
    state = G._seed_state(Q.device)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        G.gat_attention(plan, Q, K, V, H, drop_rate=0.5, drop_seed=G.new_drop_seed(Q.device))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        seed_t = G.new_drop_seed(Q.device)
        assert isinstance(seed_t, torch.Tensor)                  # under capture: a device tensor, not a frozen integer
        out = G.gat_attention(plan, Q, K, V, H, drop_rate=0.5, drop_seed=seed_t)
    outs, seeds = [], []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        outs.append(out.clone())
        seeds.append(int(seed_t.item()))
    assert len(set(seeds)) == 3 and int(state.item()) == seeds[-1]
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    for o, s in zip(outs, seeds):                                # the eager launch with the same seed BY VALUE: same mask
        assert torch.equal(o, G.gat_attention(plan, Q, K, V, H, drop_rate=0.5, drop_seed=s & 0xFFFFFFFFFFFFFFFF))
    # a whole captured training step, weights frozen (lr = 0): the loss moves between replays because the mask does, and the
    # backward of each replay regenerates ITS forward's mask (gradients finite, loss finite)
    x = tfg._lib.as_f32(x_np)
    layer = tfg.layers.GAT(8, attention_units=8, num_heads=H, edge_drop_rate=0.6)
    cache = {}
    with torch.no_grad():
        layer([x, ei], cache=cache)
    layer.trainable(True)
    opt = torch.optim.SGD(layer.parameters(), lr=0.0)
    step = tfg.CapturedTrainStep(lambda: (layer([x, ei], training=True, cache=cache) ** 2).mean(), opt)
    losses = [float(step().detach()) for _ in range(4)]
    assert len(set(losses)) == 4 and all(np.isfinite(losses))
    assert all(torch.isfinite(p.grad).all() for p in layer.parameters())


def test_edge_weights_written_in_place_are_seen(tfg, oracle):
    """The CSR-ordered copy of the caller's edge weights is memoised per cache on the array object; a torch in-place write to
    that tensor (its version counter) must invalidate it, like a new tensor does."""
    rng = np.random.Generator(np.random.PCG64(77))
    n, e, f = 300, 2400, 20
    ei = rng.integers(0, n, size=(2, e)).astype(np.int32)
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = (rng.random(e, dtype=np.float32) + 0.5)
    layer = tfg.layers.MeanGraphSage(8, activation=None)
    eid, xd, wd = torch.as_tensor(ei, device="cuda"), torch.as_tensor(x, device="cuda"), torch.as_tensor(w, device="cuda")
    cache = {}
    a = layer([xd, eid, wd], cache=cache).clone()
    flip = torch.as_tensor(rng.random(e, dtype=np.float32) + 0.25, device="cuda")
    wd.mul_(flip)                                                    # same tensor object, new contents
    b = layer([xd, eid, wd], cache=cache)
    fresh = layer([xd, eid, wd.clone()], cache={})
    assert torch.equal(b, fresh) and not torch.equal(a, b)


def test_max_pool_sage_layer0_backward_on_a_graph_without_edges(tfg):
    """The destination-major weight gradient of the pooling MLP (tfgx_pool_mlp_max_wgrad_f32) loads unconditionally inside its row
    loop: a graph with NO edge must not reach that loop — every pooled value is float32 lowest, nothing passes the ReLU, the
    MLP's gradients are exact zeros and the self half trains as usual."""
    import numpy as np
    n, f = 300, 100
    x = torch.randn(n, f, device="cuda")
    ei = np.zeros((2, 0), np.int32)
    layer = tfg.layers.MaxPoolGraphSage(256, activation=tfg.relu, concat=True)
    layer._maybe_build([x])
    layer.trainable(True)
    out = layer([x, ei, np.zeros(0, np.float32)], cache={})
    g = torch.randn(n, 256, device="cuda")
    g[:, 128:] = 0.0                      # the neighbour half pooled float32 lowest: no gradient through its overflow
    out.backward(g)
    grads = {k: v.grad for k, v in layer.weights.items()}
    assert all(v is None or bool(torch.isfinite(v).all()) for v in grads.values())
    for k in ("mlp_kernel", "mlp_bias"):
        assert grads[k] is None or float(grads[k].abs().max()) == 0.0
    assert float(grads["self_kernel"].abs().max()) > 0.0
