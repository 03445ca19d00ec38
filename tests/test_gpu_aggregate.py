# coding=utf-8
"""GPU parity: HIP gather-scale-segment-reduce (through the C ABI) vs the CPU oracle, same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _graph(oracle, n, e, f, seed=0, weighted=True):
    ei = oracle.synthetic_edges(n, e, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32) if weighted else None
    return x, ei, w


@pytest.mark.parametrize("f", [1, 3, 7, 16, 41, 64, 100, 128, 200, 256, 300, 602, 1030])
@pytest.mark.parametrize("red", ["sum", "mean", "max"])
def test_aggregate_widths(tfg, oracle, f, red):
    n, e = 500, 6000
    x, ei, w = _graph(oracle, n, e, f, seed=f)
    reducers = {"sum": (tfg.nn.sum_reducer, oracle.sum_reducer), "mean": (tfg.nn.mean_reducer, oracle.mean_reducer),
                "max": (tfg.nn.max_reducer, oracle.max_reducer)}
    g, o = reducers[red]
    got = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, g, tfg.nn.identity_updater).cpu().numpy()
    ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, o, oracle.identity_updater)
    assert_parity(got, ref, what="aggregate {} F={}".format(red, f))


@pytest.mark.parametrize("mapper", ["identity", "gcn"])
@pytest.mark.parametrize("red", ["sum", "mean", "max"])
@pytest.mark.parametrize("upd", ["sum", "identity"])
def test_aggregate_builtin_triples(tfg, oracle, mapper, red, upd):
    x, ei, w = _graph(oracle, 300, 2500, 24, seed=3)
    gm = {"identity": tfg.nn.identity_mapper, "gcn": tfg.nn.gcn_mapper}[mapper]
    om = {"identity": oracle.identity_mapper, "gcn": oracle.gcn_mapper}[mapper]
    gr = getattr(tfg.nn, red + "_reducer")
    orr = getattr(oracle, red + "_reducer")
    gu = getattr(tfg.nn, upd + "_updater")
    ou = getattr(oracle, upd + "_updater")
    got = tfg.nn.aggregate_neighbors(x, ei, w, gm, gr, gu).cpu().numpy()
    ref = oracle.aggregate_neighbors(x, ei, w, om, orr, ou)
    assert_parity(got, ref, what="{}-{}-{}".format(mapper, red, upd))


def test_empty_segments_and_isolated_nodes(tfg, oracle):
    """TF semantics: empty segment -> 0 (sum, mean), float32 lowest (max)."""
    n, f = 40, 12
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal((n, f), dtype=np.float32)
    ei = np.array([[0, 0, 3, 3, 3, 39], [1, 2, 0, 0, 5, 38]], dtype=np.int32)   # duplicate edge (3,0); most rows empty
    for red in ["sum", "mean", "max"]:
        got = tfg.nn.aggregate_neighbors(x, ei, None, tfg.nn.identity_mapper, getattr(tfg.nn, red + "_reducer"),
                                         tfg.nn.identity_updater).cpu().numpy()
        ref = oracle.aggregate_neighbors(x, ei, None, oracle.identity_mapper, getattr(oracle, red + "_reducer"),
                                         oracle.identity_updater)
        assert_parity(got, ref, what="empty-" + red)
    got = tfg.nn.aggregate_neighbors(x, ei, None, tfg.nn.identity_mapper, tfg.nn.max_reducer,
                                     tfg.nn.identity_updater).cpu().numpy()
    assert got[10, 0] == np.float32(-3.4028234663852886e38)


def test_no_edges_returns_x(tfg):
    x = np.ones((5, 3), np.float32)
    out = tfg.nn.aggregate_neighbors(x, np.zeros((0,), np.int32), None)
    assert np.array_equal(out.cpu().numpy(), x)
    out = tfg.nn.aggregate_neighbors(x, np.zeros((2, 0), np.int32), None)
    assert np.array_equal(out.cpu().numpy(), x)


def test_out_of_range_index_raises(tfg):
    x = np.ones((5, 3), np.float32)
    with pytest.raises(tfg._lib.TfgxError):
        tfg.nn.aggregate_neighbors(x, np.array([[0, 7], [1, 2]], np.int32), None)
    with pytest.raises(tfg._lib.TfgxError):
        tfg.nn.aggregate_neighbors(x, np.array([[0, 1], [1, -2]], np.int32), None)


def test_hub_row_and_ragged_degrees(tfg, oracle):
    """One destination with 20k in-edges next to degree-0/1 rows (skewed graph)."""
    n, f = 3000, 100
    rng = np.random.Generator(np.random.PCG64(9))
    x = rng.standard_normal((n, f), dtype=np.float32)
    hub_src = rng.integers(0, n, size=20000, dtype=np.int32)
    rest = oracle.synthetic_edges(n, 8000, seed=2)
    ei = np.concatenate([np.stack([np.full_like(hub_src, 17), hub_src]), rest], axis=1).astype(np.int32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
    deg = np.bincount(ei[0], minlength=n).astype(np.float64)
    for red in ["sum", "mean", "max"]:
        got = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, getattr(tfg.nn, red + "_reducer"),
                                         tfg.nn.identity_updater).cpu().numpy()
        ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, getattr(oracle, red + "_reducer"),
                                         oracle.identity_updater)
        # fp32 accumulation of d terms of rms size s carries a random-walk rounding error of about eps*d*s/2 in the
        # SUM (any fp32 implementation, TF-CPU included): the 1e-5 bar is widened by that term for the hub row only.
        extra = np.zeros((n, 1))
        if red in ("sum", "mean"):
            extra = (6e-8 * deg * 1.5 * (1.0 if red == "sum" else 1.0 / np.maximum(deg, 1)))[:, None]
            extra[deg < 512] = 0.0
        err = np.abs(got.astype(np.float64) - ref)
        bound = 1e-5 + 1e-5 * np.abs(ref) + extra
        assert (err <= bound).all(), "{}: worst excess {:.3e}".format(red, float((err - bound).max()))


def test_edge_order_permutation_invariance(tfg, oracle):
    x, ei, w = _graph(oracle, 400, 5000, 32, seed=11)
    rng = np.random.Generator(np.random.PCG64(12))
    p = rng.permutation(ei.shape[1])
    a = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater)
    b = tfg.nn.aggregate_neighbors(x, ei[:, p], w[p], tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater)
    assert_parity(a.cpu().numpy(), b.cpu().numpy(), what="permutation")
    m1 = tfg.nn.aggregate_neighbors(x, ei, None, tfg.nn.identity_mapper, tfg.nn.max_reducer, tfg.nn.identity_updater)
    m2 = tfg.nn.aggregate_neighbors(x, ei[:, p], None, tfg.nn.identity_mapper, tfg.nn.max_reducer,
                                    tfg.nn.identity_updater)
    assert np.array_equal(m1.cpu().numpy(), m2.cpu().numpy())   # max is order-independent: bit-exact


def test_deterministic_bitwise(tfg, oracle):
    x, ei, w = _graph(oracle, 2000, 60000, 100, seed=13)
    cache = {}
    a = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater, cache=cache)
    b = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater, cache=cache)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_generic_python_mapper(tfg, oracle):
    """A user mapper that is not one of the built-ins goes through gather -> mapper -> HIP reducer."""
    x, ei, w = _graph(oracle, 200, 1500, 8, seed=21)

    def mapper(repeated_x, neighbor_x, edge_weight=None):
        return (neighbor_x - repeated_x) * 0.5

    got = tfg.nn.aggregate_neighbors(x, ei, w, mapper, tfg.nn.mean_reducer, tfg.nn.sum_updater).cpu().numpy()
    ref = oracle.aggregate_neighbors(x, ei, w, lambda r, nb, edge_weight=None: (nb - r) * 0.5, oracle.mean_reducer,
                                     oracle.sum_updater)
    assert_parity(got, ref, what="generic mapper")


def test_arxiv_shaped_sum(tfg, oracle):
    """BASELINE configs[1] shape (N=170k, E=1.2M, F=128): GCN-weighted segment-sum vs float64 oracle."""
    n, e, f = 170000, 1200000, 128
    x, ei, w = _graph(oracle, n, e, f, seed=0)
    got = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater)
    ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater)
    assert_parity(got.cpu().numpy(), ref, what="arxiv-shaped segment-sum")


def test_segment_softmax_and_count(tfg, oracle):
    rng = np.random.Generator(np.random.PCG64(31))
    ids = rng.integers(0, 50, size=4000, dtype=np.int32)
    for shape in [(4000,), (4000, 8)]:
        s = (rng.standard_normal(shape) * 3).astype(np.float32)
        got = tfg.nn.segment_softmax(s, ids, 60).cpu().numpy()
        ref = oracle.segment_softmax(s, ids, 60) if len(shape) == 1 else \
            np.stack([oracle.segment_softmax(s[:, h], ids, 60) for h in range(shape[1])], axis=1)
        assert_parity(got, ref, what="segment_softmax")
    assert np.array_equal(tfg.nn.segment_count(ids, 60).cpu().numpy(), oracle.segment_count(ids, 60))


@pytest.mark.parametrize("f", [100, 36, 68, 124])
def test_split_row_layout_is_bit_identical(tfg, oracle, f):
    """SplitRows (main[n, f_main] + tail[n, F - f_main]) only changes where bytes live: same FMA chain per element."""
    import torch
    from tf_geometric_amd.plan import CsrPlan, SplitRows, segment_reduce
    L = tfg._lib
    n, e = 3000, 50000
    x, ei, w = _graph(oracle, n, e, f, seed=f)
    plan = CsrPlan.build(ei, n, n)
    xd = L.as_f32(x)
    w_csr = plan.edge_attr_to_csr(w)
    sc = torch.rand(n, device="cuda")
    bias = torch.randn(f, device="cuda")
    sp = SplitRows.from_dense(xd)
    assert sp.main.shape[1] == (f // 32) * 32 and torch.equal(torch.cat([sp.main, sp.tail], 1), xd)
    for op in (L.SUM, L.MEAN, L.MAX):
        for ww in (w_csr, None):
            a = segment_reduce(plan, xd, op, w_csr=ww, self_coef=sc, bias=bias, act=1)
            b = segment_reduce(plan, sp, op, w_csr=ww, self_coef=sc, bias=bias, act=1)
            assert torch.equal(a, b)
    ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater)
    assert_parity(segment_reduce(plan, sp, L.SUM, w_csr=w_csr).cpu().numpy(), ref, what="split rows vs oracle")
    # edge-resident tail (the tail columns of each edge's source row streamed next to col / w): still bit-identical,
    # including the self-loop term (read from the node table), hub rows, and only on the plan it was built for
    spe = SplitRows.from_dense(xd).with_edge_tail(plan)
    assert spe.edge_tail.shape == (plan.num_edges, f - (f // 32) * 32)
    for op in (L.SUM, L.MEAN, L.MAX):
        for ww in (w_csr, None):
            a = segment_reduce(plan, xd, op, w_csr=ww, self_coef=sc, bias=bias, act=1)
            b = segment_reduce(plan, spe, op, w_csr=ww, self_coef=sc, bias=bias, act=1)
            assert torch.equal(a, b)
    other = CsrPlan.build(ei[:, ::-1].copy(), n, n)
    assert torch.equal(segment_reduce(other, spe, L.SUM), segment_reduce(other, xd, L.SUM))   # falls back to the gather


def test_edge_tail_on_hub_rows(tfg, oracle):
    import torch
    from tf_geometric_amd.plan import CsrPlan, SplitRows, segment_reduce
    L = tfg._lib
    n, f = 2000, 100
    rng = np.random.Generator(np.random.PCG64(5))
    ei = oracle.synthetic_edges(n, 20000, seed=6)
    hubs = np.stack([np.full(5000, 7, np.int32), rng.integers(0, n, 5000).astype(np.int32)])
    ei = np.concatenate([ei, hubs], axis=1)
    x = rng.standard_normal((n, f), dtype=np.float32)
    plan = CsrPlan.build(ei, n, n)
    assert plan.hub_info() is not None
    xd = L.as_f32(x)
    w_csr = torch.rand(plan.num_edges, device="cuda") + 0.5
    spe = SplitRows.from_dense(xd).with_edge_tail(plan)
    assert torch.equal(segment_reduce(plan, xd, L.SUM, w_csr=w_csr), segment_reduce(plan, spe, L.SUM, w_csr=w_csr))


@pytest.mark.parametrize("op_name", ["SUM", "MEAN", "MAX"])
def test_hub_rows_with_split_layout_and_epilogue(tfg, oracle, op_name):
    """the hub finalisation must read the self-loop term of a split-row source from x_tail for the
    tail columns.  Hub rows (incl. the LAST row, where the old read ran out of bounds) x {dense, SplitRows, edge tail}
    x self_coef + bias + ReLU must all be bit-identical."""
    import torch
    from tf_geometric_amd.plan import CsrPlan, SplitRows, segment_reduce
    L = tfg._lib
    op = getattr(L, op_name)
    n, f = 1500, 100
    rng = np.random.Generator(np.random.PCG64(15))
    ei = oracle.synthetic_edges(n, 12000, seed=16)
    hubs = [np.stack([np.full(4000, r, np.int32), rng.integers(0, n, 4000).astype(np.int32)]) for r in (3, n - 1)]
    ei = np.concatenate([ei] + hubs, axis=1)
    x = rng.standard_normal((n, f), dtype=np.float32)
    plan = CsrPlan.build(ei, n, n)
    hub = plan.hub_info()
    assert hub is not None and set(hub[0].cpu().tolist()) >= {3, n - 1}
    xd = L.as_f32(x)
    tg = torch.Generator(device="cuda")
    tg.manual_seed(17)
    w_csr = torch.rand(plan.num_edges, device="cuda", generator=tg) + 0.5
    sc = torch.rand(n, device="cuda", generator=tg) + 0.25
    bias = torch.randn(f, device="cuda", generator=tg) * 0.1
    kw = dict(w_csr=w_csr, self_coef=sc, bias=bias, act=L.ACT_RELU)
    dense = segment_reduce(plan, xd, op, **kw)
    sp = SplitRows.from_dense(xd)
    assert torch.equal(dense, segment_reduce(plan, sp, op, **kw))
    assert torch.equal(dense, segment_reduce(plan, sp.with_edge_tail(plan), op, **kw))
    # and against the oracle: A@x with the self term folded in as explicit (r, r, sc_r) edges
    ew = np.empty(plan.num_edges, np.float32)
    ew[plan.perm.cpu().numpy()] = w_csr.cpu().numpy()
    ar = np.arange(n, dtype=np.int32)
    ei2 = np.concatenate([ei, np.stack([ar, ar])], axis=1)
    w2 = np.concatenate([ew, sc.cpu().numpy()])
    red = dict(SUM=oracle.sum_reducer, MEAN=oracle.sum_reducer, MAX=oracle.max_reducer)[op_name]
    ref = oracle.aggregate_neighbors(x, ei2, w2, oracle.gcn_mapper, red, oracle.identity_updater)
    if op_name == "MEAN":       # the kernel's divisor is the in-degree WITHOUT the implicit self edge
        ref = ref / np.maximum(np.bincount(ei[0], minlength=n), 1)[:, None]
    ref = np.maximum(ref + bias.cpu().numpy(), 0)
    deg = np.bincount(ei[0], minlength=n).astype(np.float64)
    got = dense.cpu().numpy()
    band = 1e-5 + 1e-5 * np.abs(ref) + 6e-8 * np.sqrt(deg)[:, None] * 8        # fp32 random-walk term on 4000-term rows
    assert (np.abs(got - ref) <= band).all()


def test_neighbor_count_mapper_and_utils(tfg, oracle):
    x, ei, w = _graph(oracle, 120, 900, 5, seed=41)
    got = tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.neighbor_count_mapper, tfg.nn.sum_reducer,
                                     tfg.nn.identity_updater).cpu().numpy()
    assert np.array_equal(got[:, 0], np.bincount(ei[0], minlength=120).astype(np.float32))
    ei2, w2 = tfg.utils.add_self_loop_edge(ei, 120, w, fill_weight=2.0)
    oe, ow = oracle.add_self_loop_edge(ei, 120, w, fill_weight=2.0)
    assert np.array_equal(ei2, oe) and np.array_equal(w2, ow)
    ei3, w3 = tfg.utils.add_self_loop_edge(tfg._lib.as_i32(ei), 120)
    assert np.array_equal(ei3.cpu().numpy(), oe) and w3 is None


def _np_merge(ei, props, modes):
    """tf.unique-order merge restated with numpy (first-occurrence order)."""
    n = int(ei.max()) + 1
    h = ei[0].astype(np.int64) * n + ei[1]
    _, first, inv = np.unique(h, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    uidx = rank[inv]
    u = ei[:, np.sort(first)]
    out = []
    for p, m in zip(props, modes):
        res = np.zeros(order.size, np.float64) if m in ("sum", "mean") else np.full(order.size, -np.inf if m == "max" else np.inf)
        cnt = np.bincount(uidx, minlength=order.size)
        if m in ("sum", "mean"):
            np.add.at(res, uidx, p.astype(np.float64))
            res = res / cnt if m == "mean" else res
        elif m == "max":
            np.maximum.at(res, uidx, p)
        else:
            np.minimum.at(res, uidx, p)
        out.append(res.astype(np.float32))
    return u, out


def test_edge_preprocessing_on_device(tfg):
    # the reference's docstring example (graph_utils.py:157-158)
    d, _ = tfg.utils.convert_edge_to_directed(np.array([[1, 3, 5], [2, 1, 4]], np.int32))
    # the reference's answer (run unmodified, tests/golden/reference_cases.npz edge_preprocessing::doc_directed):
    # convert_edge_to_upper's first-occurrence order through tf.unique, then the mirrored copies
    assert d.tolist() == [[1, 1, 4, 2, 3, 5], [2, 3, 5, 1, 1, 4]]
    rng = np.random.Generator(np.random.PCG64(8))
    ei = rng.integers(0, 60, size=(2, 3000), dtype=np.int32)              # many duplicates and self-loops
    w = rng.uniform(0.5, 1.5, 3000).astype(np.float32)
    u, props = tfg.utils.merge_duplicated_edge(ei, [w, w, w, w], ["sum", "mean", "max", "min"])
    ru, rprops = _np_merge(ei, [w, w, w, w], ["sum", "mean", "max", "min"])
    assert np.array_equal(u, ru)
    for a, b in zip(props, rprops):
        assert_parity(a, b, what="merged edge prop")
    up, (uw,) = tfg.utils.convert_edge_to_upper(ei, [w])
    rup, (ruw,) = _np_merge(np.stack([ei.min(0), ei.max(0)]), [w], ["sum"])
    assert np.array_equal(up, rup)
    assert_parity(uw, ruw, what="upper weights")
    de, (dw,) = tfg.utils.convert_edge_to_directed(ei, [w])
    m = rup[0] != rup[1]
    assert np.array_equal(de, np.concatenate([rup, rup[::-1][:, m]], axis=1))
    assert_parity(dw, np.concatenate([ruw, ruw[m]]), what="directed weights")
    e2, w2 = tfg.utils.remove_self_loop_edge(ei, w)
    keep = ei[0] != ei[1]
    assert np.array_equal(e2, ei[:, keep]) and np.array_equal(w2, w[keep])


def test_graph_readout_pools(tfg, oracle):
    """nn/pool/common_pool.py:7-52 on the segment kernels (graph id = destination)."""
    rng = np.random.Generator(np.random.PCG64(17))
    x = rng.standard_normal((900, 11), dtype=np.float32)
    gid = np.sort(rng.integers(0, 37, size=900, dtype=np.int32))
    gid[gid == 20] = 21                                              # graph 20 is empty
    cnt = np.bincount(gid, minlength=40).astype(np.float32)
    s = oracle.unsorted_segment_sum(x, gid, 40)
    assert_parity(tfg.nn.sum_pool(x, gid, 40).cpu().numpy(), s, what="sum_pool")
    assert_parity(tfg.nn.mean_pool(x, gid, 40).cpu().numpy(), s / (cnt[:, None] + 1e-8), what="mean_pool")
    assert np.array_equal(tfg.nn.max_pool(x, gid, 40).cpu().numpy(), oracle.unsorted_segment_max(x, gid, 40))
    assert np.array_equal(tfg.nn.min_pool(x, gid, 40).cpu().numpy(), -oracle.unsorted_segment_max(-x, gid, 40))
    assert tfg.nn.max_pool(x, gid).shape[0] == int(gid.max()) + 1
    for cls, fn in ((tfg.layers.MeanPool, tfg.nn.mean_pool), (tfg.layers.SumPool, tfg.nn.sum_pool),
                    (tfg.layers.MaxPool, tfg.nn.max_pool), (tfg.layers.MinPool, tfg.nn.min_pool)):
        import torch
        assert torch.equal(cls()([x, gid, 40]), fn(x, gid, 40)) and cls()([x, gid]).shape[0] == int(gid.max()) + 1


def test_sampler_output_carries_a_ready_plan(tfg, oracle):
    """Tensor-mode sampler output is already grouped by destination: the layers reuse its CSR (no second sort) and get
    the same result as from the bare edge list; gradients flow through the attached plan's transposed view."""
    import torch
    from tf_geometric_amd.plan import CsrPlan
    n = 500
    ei = oracle.synthetic_edges(n, 9000, seed=21)
    ei = ei[:, ei[0] < 470]                                            # the last 30 nodes receive nothing
    rng = np.random.Generator(np.random.PCG64(2))
    x = rng.standard_normal((n, 12), dtype=np.float32)
    sampler = tfg.utils.RandomNeighborSampler(tfg._lib.as_i32(ei))
    sei, sw = sampler.sample(k=7, seed=3)
    assert sei.is_cuda and hasattr(sei, "_tfgx_plan")
    plan = CsrPlan.from_cache(sei, n, n, None)
    assert plan.n_dst == n and plan.num_edges == sei.shape[1] and bool((plan.perm == torch.arange(plan.num_edges, device="cuda")).all())
    layer = tfg.layers.MeanGraphSage(8, activation=tfg.relu)
    a = layer([x, sei, sw])
    b = layer([x, sei.cpu().numpy(), sw.cpu().numpy()])               # numpy copy: no attached plan, sorted again
    assert torch.equal(a, b)
    layer.trainable(True)
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    layer([xt, sei, sw]).square().sum().backward()
    g1 = xt.grad.clone()
    xt.grad = None
    layer([xt, sei.cpu().numpy(), sw.cpu().numpy()]).square().sum().backward()
    assert torch.allclose(g1, xt.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("as_tuple", [False, True])
def test_neighbor_sampler_subgraph(tfg, oracle, as_tuple):
    """sampled_node_index (graph_utils.py:689-731): rows in the given order (with a duplicate and an id without edges),
    neighbours restricted to the column set and renamed to virtual ids.  sample_all is deterministic -> exact match
    with the restated loop; k / ratio / padding -> counts, subset, no repeats."""
    n = 400
    ei = oracle.synthetic_edges(n, 6000, seed=9)
    ei = ei[:, ei[0] != 11]                                             # node 11 has no in-edges
    w = np.arange(ei.shape[1], dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(3))
    rows = rng.permutation(n)[:150].astype(np.int32)
    rows[5], rows[9] = 11, rows[2]                                      # an empty row, a duplicate row
    cols = rng.permutation(n)[:220].astype(np.int32)
    sni = (rows, cols) if as_tuple else rows
    sampler = tfg.utils.RandomNeighborSampler(ei, w)
    ref_rows, ref = oracle.neighbor_lists(ei, w, sni)
    sei, sw = sampler.sample(sampled_node_index=sni)                    # sample_all
    ref_ei = np.concatenate([np.stack([np.full(len(ids), r), ids]) for r, (ids, _) in zip(ref_rows, ref)], axis=1)
    ref_w = np.concatenate([ws for _, ws in ref])
    assert np.array_equal(sei, ref_ei) and np.array_equal(sw, ref_w)
    avail = {r: set(zip(ids.tolist(), ws.tolist())) for r, (ids, ws) in zip(ref_rows, ref)}
    for kw in (dict(k=2), dict(ratio=0.5), dict(k=6, padding=True)):
        sei, sw = sampler.sample(sampled_node_index=sni, seed=5, **kw)
        cnt = np.bincount(sei[0], minlength=len(rows))
        for r in range(len(rows)):
            d = len(avail.get(r, ()))
            want = (min(d, 2) if "k" in kw and not kw.get("padding") else
                    (int(np.ceil(d * 0.5)) if "ratio" in kw else (6 if d > 0 else 0)))
            assert cnt[r] == want
            got = list(zip(sei[1][sei[0] == r].tolist(), sw[sei[0] == r].tolist()))
            assert set(got) <= avail.get(r, set())
            if not kw.get("padding"):
                assert len(set(got)) == len(got)


@pytest.mark.parametrize("n,num_src,kw", [(5000, 97, dict(k=3)), (5000, 97, dict(ratio=0.3)), (20000, 3, dict(k=50)),
                                          (300, 400, dict(ratio=1.0)), (1, 1, dict(k=1)), (777, 50, dict(k=0)),
                                          (4000, 60, dict(ratio=2.5)), (200000, 5000, dict(ratio=0.5))])
def test_topk_pool(tfg, oracle, n, num_src, kw):
    """nn/pool/topk_pool.py:6-87: bit-exact index lists — unsorted sources, gaps in the ids, heavy ties (quantised
    scores, +-0.0), k = 0, ratio beyond 1."""
    rng = np.random.Generator(np.random.PCG64(n + num_src))
    src = rng.integers(0, num_src, size=n).astype(np.int32)
    src[src == 7] = 8 if num_src > 8 else src[src == 7]            # a gap in the ids
    score = np.round(rng.standard_normal(n).astype(np.float32) * 4) / 4   # many equal scores
    score[rng.random(n) < 0.05] = -0.0
    got = tfg.nn.topk_pool(src, score, **kw)
    ref = oracle.topk_pool(src, score, **kw)
    assert got.dtype == np.int32 and np.array_equal(got, ref)
    import torch
    got_t = tfg.nn.topk_pool(torch.from_numpy(src).cuda(), torch.from_numpy(score).cuda(), **kw)
    assert got_t.is_cuda and np.array_equal(got_t.cpu().numpy(), ref)


def test_topk_pool_errors(tfg):
    src, score = np.zeros(4, np.int32), np.ones(4, np.float32)
    with pytest.raises(Exception, match="either k or ratio"):
        tfg.nn.topk_pool(src, score)
    with pytest.raises(Exception, match="not both"):
        tfg.nn.topk_pool(src, score, k=1, ratio=0.5)
    assert tfg.nn.topk_pool(np.zeros(0, np.int32), np.zeros(0, np.float32), k=2).shape == (0,)
    from tf_geometric_amd import _lib as L
    import torch
    lib = L.require_gpu()
    seg = torch.tensor([0, 5, 1], dtype=torch.int32, device="cuda")      # id 5 outside [0, 3)
    sc = torch.ones(3, device="cuda")
    out, cnt = torch.empty(3, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    nb = lib.tfgx_segment_topk_workspace_bytes(3, 3)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rc = lib.tfgx_segment_topk(L.ptr(seg), L.ptr(sc), 3, 3, 1, 0.0, L.ptr(out), L.ptr(cnt), L.ptr(ws), nb, L.stream_ptr())
    assert rc == L.ERR_INDEX if hasattr(L, "ERR_INDEX") else rc != 0
    rc = lib.tfgx_segment_topk(L.ptr(seg), L.ptr(sc), 3, 6, 1, 0.0, L.ptr(out), L.ptr(cnt), L.ptr(ws), 16, L.stream_ptr())
    assert rc != 0 and b"workspace" in lib.tfgx_last_error()


def test_random_neighbor_sampler(tfg, oracle):
    """Distributional parity with RandomNeighborSampler.sample (graph_utils.py:667-772): counts, subset, no repeats,
    order of rows, determinism per seed, and uniformity of the without-replacement draw."""
    n = 300
    ei = oracle.synthetic_edges(n, 12000, seed=4)
    ei = ei[:, ei[0] != 5]                                              # node 5 has no in-edges
    w = np.arange(ei.shape[1], dtype=np.float32)                        # unique weights identify the edge
    sampler = tfg.utils.RandomNeighborSampler(ei, w)
    deg = np.bincount(ei[0], minlength=n)
    nbrs = {r: set(zip(ei[1][ei[0] == r].tolist(), w[ei[0] == r].tolist())) for r in range(n)}
    for k in (1, 5, 25):
        sei, sw = sampler.sample(k=k, seed=7)
        cnt = np.bincount(sei[0], minlength=n)
        assert np.array_equal(cnt, np.minimum(deg, k)) and (np.diff(sei[0]) >= 0).all()
        for r in (0, 17, 123):
            got = list(zip(sei[1][sei[0] == r].tolist(), sw[sei[0] == r].tolist()))
            assert len(set(got)) == len(got) and set(got) <= nbrs[r]
        sei2, sw2 = sampler.sample(k=k, seed=7)
        assert np.array_equal(sei, sei2) and np.array_equal(sw, sw2)
        sei3, _ = sampler.sample(k=k, seed=8)
        assert not np.array_equal(sei, sei3) or k >= deg.max()
    alle, allw = sampler.sample()
    assert alle.shape[1] == ei.shape[1] and np.array_equal(np.sort(allw), np.sort(w))
    re_, _ = sampler.sample(ratio=0.5)
    assert np.array_equal(np.bincount(re_[0], minlength=n), np.ceil(deg * 0.5).astype(np.int64))
    pe, _ = sampler.sample(k=60, padding=True)
    assert np.array_equal(np.bincount(pe[0], minlength=n), np.where(deg > 0, 60, 0))
    # uniformity: a row with d neighbours, k = 1, many seeds -> each neighbour about equally often
    r = int(np.argmax(deg))
    hits = {}
    for seed in range(400):
        e1, w1 = sampler.sample(k=1, seed=seed)
        hits[w1[e1[0] == r][0]] = hits.get(w1[e1[0] == r][0], 0) + 1
    assert len(hits) > 0.8 * min(deg[r], 400 * 0.63) and max(hits.values()) <= 400 / deg[r] * 6 + 6


def test_sampler_fresh_seed_and_hub_rows(tfg):
    """(i) the reference draws fresh np.random samples on every call — sample() without a seed must not
    return the same sub-graph twice, yet be reproducible under torch.manual_seed; (ii) sample() / ratio= sampling must
    work on rows with more than 256 neighbours (keep-all needs no scratch; large draws use selection sampling)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(3))
    n, hub_deg = 50, 3000
    hub = np.stack([np.full(hub_deg, 7, np.int32), rng.integers(0, n, hub_deg).astype(np.int32)])
    rest = rng.integers(0, n, size=(2, 600)).astype(np.int32)
    ei = np.concatenate([hub, rest], axis=1)
    w = np.arange(ei.shape[1], dtype=np.float32)
    s = tfg.utils.RandomNeighborSampler(ei, w)
    deg = np.bincount(ei[0], minlength=n)
    alle, allw = s.sample()                                           # sample_all on a hub row
    assert alle.shape[1] == ei.shape[1] and np.array_equal(np.sort(allw), np.sort(w))
    re_, rw = s.sample(ratio=0.5, seed=1)                             # 1500 draws out of 3000: selection sampling
    assert np.array_equal(np.bincount(re_[0], minlength=n), np.ceil(deg * 0.5).astype(np.int64))
    hub_w = rw[re_[0] == 7]
    assert len(set(hub_w.tolist())) == hub_w.size and set(hub_w.tolist()) <= set(w[ei[0] == 7].tolist())
    assert (np.diff(hub_w) > 0).all()                                 # neighbour order kept
    first_half = float((hub_w < np.median(w[ei[0] == 7])).mean())     # uniform over the row, not front-loaded
    assert 0.42 < first_half < 0.58
    # wave-per-row path (rows with more than 64 neighbours): every neighbour is included with probability m / d — also
    # when the 64 strata are small (d = 100 .. 3000, m = 40 .. 80 % of d), where a fixed per-stratum quota would make
    # some positions certain and others impossible
    for d_row, ratio in ((100, 0.5), (257, 0.4), (3000, 0.8)):
        e2 = np.stack([np.zeros(d_row, np.int32), np.arange(d_row, dtype=np.int32) % n])
        s2 = tfg.utils.RandomNeighborSampler(e2, np.arange(d_row, dtype=np.float32))
        m_row = int(np.ceil(d_row * ratio))
        freq = np.zeros(d_row)
        trials = 300
        for seed in range(trials):
            ee, ww = s2.sample(ratio=ratio, seed=1000 + seed)
            assert ww.size == m_row and len(set(ww.tolist())) == m_row and (np.diff(ww) > 0).all()
            freq[ww.astype(np.int64)] += 1
        p = m_row / d_row
        sigma = np.sqrt(p * (1 - p) / trials)
        assert np.abs(freq / trials - p).max() < 5.5 * sigma + 1e-9, (d_row, ratio, np.abs(freq / trials - p).max(), sigma)
    a, _ = s.sample(k=3)
    b, _ = s.sample(k=3)
    assert not np.array_equal(a, b)
    torch.manual_seed(11)
    c, _ = s.sample(k=3)
    torch.manual_seed(11)
    d, _ = s.sample(k=3)
    assert np.array_equal(c, d)


@pytest.mark.parametrize("heads", [1, 3, 8])
def test_segment_softmax_hub_rows(tfg, oracle, heads):
    """Standalone segment softmax on a graph with a 9000-edge segment: cooperative lanes for ordinary rows, the chunked
    path (per-chunk statistics, ordered fold, per-chunk normalisation) for hub rows — with the plan's own policy (the
    workgroup-per-row kernel when the plan declares no hub) and with a forced low threshold — equals the oracle."""
    import tf_geometric_amd.plan as P
    rng = np.random.Generator(np.random.PCG64(5 + heads))
    n = 300
    ids = np.concatenate([np.full(9000, 4, np.int32), rng.integers(0, n, 5000).astype(np.int32),
                          np.full(1500, 250, np.int32)])
    ids = ids[rng.permutation(ids.size)]
    data = (rng.standard_normal((ids.size, heads)) * 3).astype(np.float32)
    ref = np.stack([oracle.segment_softmax(data[:, h], ids, n) for h in range(heads)], axis=1)
    old = (P.HUB_THRESHOLD, P.HUB_CHUNK)
    try:
        for thr in (None, 128):
            P.HUB_THRESHOLD, P.HUB_CHUNK = thr, (None if thr is None else 96)
            got = tfg.nn.segment_softmax(data if heads > 1 else data[:, 0], ids, n).cpu().numpy().reshape(ids.size, heads)
            assert_parity(got, ref, tol=2e-6, what="segment_softmax hub thr={}".format(thr))
    finally:
        P.HUB_THRESHOLD, P.HUB_CHUNK = old


def test_segment_op_with_pad_routes_gradient_and_refusals(tfg, oracle):
    """nn/kernel/segment.py:5-23.  This module's own sorted segment ops take ONE launch on the unsorted rows; any other
    callable is run as the reference runs it (sort, gather, op, pad) — same values; gradients flow through the kernel's own
    backward; ids >= num_segments and unsorted ids handed to a sorted op are refused."""
    import functools
    seg = tfg.nn.kernel.segment
    rng = np.random.Generator(np.random.PCG64(91))
    ids = rng.integers(0, 50, size=2000).astype(np.int32)
    ids[ids == 13] = 14
    x = rng.standard_normal((2000, 7), dtype=np.float32)
    for kind in ("sum", "mean", "max", "min"):
        op = getattr(seg, "segment_" + kind)
        fast = seg.segment_op_with_pad(op, x, ids, 60)
        slow = seg.segment_op_with_pad(lambda d, i, op=op: op(d, i), x, ids, 60)        # a plain callable: the reference's route
        ref = oracle.segment_op_with_pad(functools.partial(oracle.sorted_segment, kind), x, ids, 60)
        assert tuple(fast.shape) == (60, 7)
        assert_parity(fast.cpu().numpy(), ref, what="segment_op_with_pad " + kind)
        if kind in ("max", "min"):
            assert np.array_equal(fast.cpu().numpy(), ref) and torch.equal(fast, slow)
        else:
            assert_parity(slow.cpu().numpy(), ref, what="segment_op_with_pad generic " + kind)
    xd = torch.tensor(x, device="cuda", requires_grad=True)
    seg.segment_op_with_pad(seg.segment_sum, xd, ids, 60).sum().backward()
    assert torch.equal(xd.grad, torch.ones_like(xd))
    xd.grad = None
    seg.segment_op_with_pad(seg.segment_max, xd, ids, 60)[:, 0].sum().backward()
    g0 = xd.grad[:, 0].cpu().numpy()
    assert g0.sum() == len(np.unique(ids)) and (xd.grad[:, 1:] == 0).all()
    with pytest.raises(ValueError):
        seg.segment_op_with_pad(seg.segment_sum, x, ids, 49)
    with pytest.raises(ValueError):
        seg.segment_max(x, ids)                                                          # ids not ascending
    assert tuple(seg.segment_max(x[:0], ids[:0]).shape) == (0, 7)
    # integer data keeps its dtype (segment.py:12-18 pads with dtype=reduced_data.dtype) and stays exact, or is refused
    counts = rng.integers(0, 1000, size=(2000, 3)).astype(np.int64)
    got = seg.segment_op_with_pad(seg.segment_sum, counts, ids, 60)
    want = np.zeros((60, 3), dtype=np.int64)
    np.add.at(want, ids, counts)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want)
    assert seg.segment_max(np.sort(ids), np.sort(ids)).dtype == torch.int32
    with pytest.raises(TypeError):
        seg.segment_op_with_pad(seg.segment_sum, counts * 100000, ids, 60)               # sums beyond 2^24: not exact in float32
    with pytest.raises(TypeError):
        seg.segment_op_with_pad(seg.segment_mean, counts, ids, 60)
    bad = ids.copy()
    bad[5] = -1
    with pytest.raises(ValueError):
        seg.segment_op_with_pad(seg.segment_sum, x, bad, 60)                             # negative id: TF raises InvalidArgumentError


@pytest.mark.parametrize("f", [128, 160, 192, 256, 384, 512, 1024])
def test_wide_rows_in_column_blocks_are_bit_identical_to_one_burst(tfg, oracle, f):
    """Wide rows made of whole 128-byte lines are gathered in 64-column blocks on grid.y (128 columns at F = 256), 16 pieces in
    flight per lane, the last partial batch as one masked batch: every output element keeps its own in-order FMA chain, so
    the launch equals the one-burst-per-row launch BIT FOR BIT (sum, mean with the self-loop term and an epilogue, max) and
    sits in the oracle's band; a table whose rows are not line-aligned keeps the burst."""
    from tf_geometric_amd import plan as P
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(f))
    n, e = 1500, 90000
    ei = oracle.synthetic_edges(n, e, seed=f)
    ei = ei[:, (ei[0] != 5) & (ei[0] != 1499)]                      # empty rows: the first block and the last
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    plan = P.CsrPlan.build(L.as_i32(ei), n, n)
    xd, wd = L.as_f32(x), plan.edge_attr_to_csr(L.as_f32(w))
    sc = L.as_f32(rng.uniform(0.1, 0.9, n).astype(np.float32))
    bias = L.as_f32(rng.standard_normal(f).astype(np.float32))
    name = P.segment_reduce(plan, xd, L.SUM, w_csr=wd, describe=True)
    assert name.endswith(", 16>") and ("<4, 16, 1," in name or (f == 256 and "<4, 32, 1," in name)), name
    assert P.segment_reduce(plan, xd, L.SUM, w_csr=wd, describe=True, wide_blocks=-1).endswith(", 0>")
    for op, kw in ((L.SUM, dict(w_csr=wd, self_coef=sc, bias=bias, act=L.ACT_RELU)), (L.MEAN, dict(w_csr=wd)), (L.MAX, dict())):
        blocks = P.segment_reduce(plan, xd, op, wide_blocks=1, **kw)
        burst = P.segment_reduce(plan, xd, op, wide_blocks=-1, **kw)
        assert torch.equal(blocks, burst)
    got = P.segment_reduce(plan, xd, L.SUM, w_csr=wd).cpu().numpy()
    ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.identity_updater)
    assert_parity(got, ref, what="wide rows F={}".format(f))
    gotm = P.segment_reduce(plan, xd, L.MAX).cpu().numpy()
    refm = oracle.aggregate_neighbors(x, ei, None, oracle.identity_mapper, oracle.max_reducer, oracle.identity_updater)
    assert np.array_equal(gotm, refm)
    # rows 4 floats further apart are no longer whole lines: one burst per row, same values
    wide = torch.empty((n, f + 4), device="cuda")
    wide[:, :f] = xd
    assert P.segment_reduce(plan, wide[:, :f], L.SUM, w_csr=wd, describe=True).endswith(", 0>")
    assert torch.equal(P.segment_reduce(plan, wide[:, :f], L.SUM, w_csr=wd), P.segment_reduce(plan, xd, L.SUM, w_csr=wd, wide_blocks=-1))


def test_power_of_two_row_strides_are_avoided(tfg):
    """plan.gather_friendly_ld never hands out a power-of-two row stride of 512 bytes or more (hot rows of a power-law graph,
    and every row at 2 KB, fold onto the same memory channels: profiles/r05_ab_ld_pad.jsonl), and keeps the strides that were
    already line-friendly."""
    from tf_geometric_amd import plan as P
    assert [P.gather_friendly_ld(v) for v in (20, 47, 64, 100, 128, 160, 256, 512, 1024)] == [32, 48, 64, 100, 160, 160, 288, 544, 1056]
    t = P.gather_friendly_empty(10, 256, "cuda")
    assert tuple(t.shape) == (10, 256) and t.stride(0) == 288


def test_relaid_copy_of_a_power_of_two_stride_table_is_memoised_and_fails_safe(tfg):
    """VERDICT r5 item 7 / ADVICE: a caller's table with a 2 KB row stride is gathered from a copy whose rows sit 128 bytes
    further apart (plan.relaid_for_gather); the copy used to be made on EVERY call.  It is memoised per table now: the second
    call with the same tensor copies nothing, a torch-visible write re-copies, and a write behind the version counter is
    caught by the sampled-row comparison the promoted layouts use — the call returns the fresh aggregation."""
    import torch
    from tf_geometric_amd import _lib as L, plan as P, synthetic
    n, e, F = 300000, 4000000, 512            # 614 MB table: beyond the 512 MB floor of the re-layout
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=4))
    plan = P.CsrPlan.build(ei, n, n)
    x = torch.randn(n, F, device="cuda")
    P.release_relaid_copies()
    st = lambda k: P.RELAY_STATS.get(k, 0)                             # noqa: E731
    copies, hits, caught = st("copies"), st("hits"), st("stale_copies_caught")
    o1 = P.segment_reduce(plan, x, L.SUM)
    assert st("copies") == copies + 1
    o2 = P.segment_reduce(plan, x, L.SUM)
    assert st("copies") == copies + 1 and st("hits") == hits + 1 and torch.equal(o1, o2)          # no second copy
    P.RELAY_POW2_TABLES = False
    try:
        assert torch.equal(P.segment_reduce(plan, x, L.SUM), o1)       # same bits as the table itself
    finally:
        P.RELAY_POW2_TABLES = True
    x.mul_(2.0)                                                         # torch-visible write: copied again
    o3 = P.segment_reduce(plan, x, L.SUM)
    assert st("copies") == copies + 2 and torch.equal(o3, o1 * 2.0)
    v = x._version
    x.data.mul_(0.5)                                                    # behind the version counter
    assert x._version == v
    o4 = P.segment_reduce(plan, x, L.SUM)
    assert torch.equal(o4, o1) and st("stale_copies_caught") == caught + 1 and st("copies") == copies + 3
    # inside a hipGraph capture nothing is served from the memo: a replay re-copies, so it sees the table change
    out = torch.empty_like(o1)
    cap = tfg.CapturedForward(lambda: P.segment_reduce(plan, x, L.SUM, out=out))
    assert torch.equal(cap(), o1)
    x.copy_(x * 2.0)                                                    # (a power of two: exact)
    assert torch.equal(cap(), o1 * 2.0)
    P.release_relaid_copies()
