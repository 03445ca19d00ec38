# coding=utf-8
"""GPU parity of the layer API (GCN / GAT / GraphSAGE) and the MFMA GEMM vs the CPU oracle."""
import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _graph(oracle, n, e, f, seed=0):
    ei = oracle.synthetic_edges(n, e, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
    return x, ei, w, rng


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (5, 3, 2), (127, 16, 7), (128, 100, 64), (1000, 100, 256),
                                   (333, 1433, 16), (2708, 602, 41), (4096, 128, 128), (513, 17, 129)])
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_bias_act(tfg, oracle, m, k, n, relu):
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(m * 7 + k))
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = oracle.glorot_uniform(rng, k, n)
    bias = rng.standard_normal(n).astype(np.float32) * 0.1
    got = gemm_bias_act(a, b, bias=bias, act=1 if relu else 0).cpu().numpy()
    ref = oracle.matmul(a, b) + bias
    if relu:
        ref = np.maximum(ref, 0)
    assert_parity(got, ref, what="gemm {}x{}x{}".format(m, k, n))


def test_gemm_transpose_detecting(tfg):
    """A = I with an asymmetric B catches a swapped C layout (cdna guide §3)."""
    from tf_geometric_amd.plan import gemm_bias_act
    n = 96
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) / 97.0
    got = gemm_bias_act(np.eye(n, dtype=np.float32), b).cpu().numpy()
    assert np.array_equal(got, b)


@pytest.mark.parametrize("cfg", [dict(), dict(renorm=False), dict(improved=True), dict(norm="left"),
                                 dict(norm="right"), dict(sym=False), dict(add_self_loop=False),
                                 dict(norm="left", add_self_loop=False)])
def test_gcn_layer(tfg, oracle, cfg):
    x, ei, w, rng = _graph(oracle, 600, 5000, 50, seed=4)
    kernel = oracle.glorot_uniform(rng, 50, 24)
    bias = (rng.standard_normal(24) * 0.1).astype(np.float32)
    layer = tfg.layers.GCN(24, activation=tfg.relu, **cfg)
    layer._maybe_build([x])
    layer.set_weights(kernel=kernel, bias=bias)
    cache = {}
    got = layer([x, ei, w], cache=cache).cpu().numpy()
    ref = oracle.gcn(x, ei, w, kernel, bias, "relu", **cfg)
    assert_parity(got, ref, what="GCN {}".format(cfg))
    got2 = layer([x, ei, w], cache=cache).cpu().numpy()     # cached plan + cached normalised adjacency
    assert np.array_equal(got, got2)


def test_gcn_no_kernel_unweighted_and_path_graph_kat(tfg, oracle):
    """3-node path graph: normalised weights are exactly 1/sqrt(6), 1/2, 1/3 (hand-derived)."""
    ei = np.array([[0, 1, 1, 2], [1, 0, 2, 1]], np.int32)
    x = np.eye(3, dtype=np.float32)
    layer = tfg.layers.GCN(3, use_kernel=False, use_bias=False)
    got = layer([x, ei]).cpu().numpy()
    s = np.float32(1.0 / np.sqrt(6.0))
    kat = np.array([[0.5, s, 0], [s, 1.0 / 3.0, s], [0, s, 0.5]], dtype=np.float32)
    assert_parity(got, kat, tol=1e-6, what="path-graph KAT")


@pytest.mark.parametrize("heads,att,units,split", [(1, 8, 8, True), (8, 8, 64, True), (8, 64, 64, True),
                                                  (4, 16, 20, True), (2, 6, 10, False), (8, 8, 16, False),
                                                  (1, 1, 41, True), (3, 9, 9, True), (8, 256, 64, True), (4, 12, 64, True),
                                                  (2, 2, 82, True), (4, 4, 10, False), (5, 20, 40, True), (2, 80, 8, True)])
def test_gat_layer(tfg, oracle, heads, att, units, split):
    x, ei, w, rng = _graph(oracle, 400, 4000, 30, seed=heads + att)
    layer = tfg.layers.GAT(units, attention_units=att, activation=tfg.relu, num_heads=heads, split_value_heads=split)
    layer._maybe_build([x])
    wq, wk = oracle.glorot_uniform(rng, 30, att), oracle.glorot_uniform(rng, 30, att)
    bq, bk = (rng.standard_normal(att) * 0.2).astype(np.float32), (rng.standard_normal(att) * 0.2).astype(np.float32)
    wv = oracle.glorot_uniform(rng, 30, units if split else units * heads)
    b = (rng.standard_normal(units) * 0.1).astype(np.float32)
    layer.set_weights(query_kernel=wq, query_bias=bq, key_kernel=wk, key_bias=bk, kernel=wv, bias=b)
    got = layer([x, ei]).cpu().numpy()
    ref = oracle.gat(x, ei, wq, bq, "relu", wk, bk, "relu", wv, b, "relu", num_heads=heads, split_value_heads=split)
    assert_parity(got, ref, what="GAT H={} A={} U={} split={}".format(heads, att, units, split))


@pytest.mark.parametrize("heads,att,units,kb", [(8, 8, 64, 4), (8, 64, 64, 3), (1, 1, 44, 2), (4, 16, 20, 7), (2, 6, 10, 5)])
def test_gat_source_blocks_equal_the_oracle(tfg, oracle, heads, att, units, kb):
    """Dense graphs run the fused attention as KB chained launches over SOURCE BLOCKS (each launch gathers K / V rows of one
    block; the raw online-softmax state is handed from launch to launch, the last one appends the self-loop edge): the same
    layer output as the reference within the plain band — only the order in which a row's edges enter its softmax sums
    changes — for every head geometry class, with empty (row, block) spans, isolated nodes and explicit self-loops."""
    from tf_geometric_amd.nn.conv import gat as G
    x, ei, w, rng = _graph(oracle, 600, 30000, 30, seed=heads + att + kb)
    ei = ei[:, (ei[0] != 17) & (ei[0] != 333)]                                           # two rows without in-edges
    ei = np.concatenate([ei, np.stack([np.arange(20, dtype=np.int32)] * 2)], axis=1)     # explicit self-loops are kept
    layer = tfg.layers.GAT(units, attention_units=att, activation=tfg.relu, num_heads=heads)
    layer._maybe_build([x])
    wq, wk = oracle.glorot_uniform(rng, 30, att), oracle.glorot_uniform(rng, 30, att)
    bq, bk = (rng.standard_normal(att) * 0.2).astype(np.float32), (rng.standard_normal(att) * 0.2).astype(np.float32)
    wv, b = oracle.glorot_uniform(rng, 30, units), (rng.standard_normal(units) * 0.1).astype(np.float32)
    layer.set_weights(query_kernel=wq, query_bias=bq, key_kernel=wk, key_bias=bk, kernel=wv, bias=b)
    ref = oracle.gat(x, ei, wq, bq, "relu", wk, bk, "relu", wv, b, "relu", num_heads=heads)
    one = layer([x, ei]).cpu().numpy()
    before = G.SOURCE_BLOCK_STATS["launches"]
    G.SOURCE_BLOCKS = kb
    try:
        got = layer([x, ei], cache={}).cpu().numpy()
    finally:
        G.SOURCE_BLOCKS = None
    assert G.SOURCE_BLOCK_STATS["launches"] == before + kb
    assert_parity(got, ref, what="GAT in {} source blocks H={} A={} U={}".format(kb, heads, att, units))
    assert_parity(got, one, what="source blocks vs one pass")
    # the policy: Reddit-like density and table size -> blocks; products-like density -> one pass
    class _P(object):
        def __init__(self, n, e):
            self.n_dst, self.n_src, self.num_edges = n, n, e
    assert G.source_block_count(_P(233000, 114000000), 8, 64) == 11 and G.source_block_count(_P(2400000, 123000000), 8, 64) == 1
    assert G.source_block_count(_P(600, 30000), 8, 64) == 1


def test_gat_large_scores_online_softmax(tfg, oracle):
    """Scores spanning about +-25 force many running-max rescales; existing self-loops are kept (duplicates).
    A score s carries an fp32 rounding error of ~|s|*1e-7 which exp() turns into the same RELATIVE error of the
    attention weight, so the bar is 1e-5 * max|s| here (true of any fp32 implementation of gat.py:79-84)."""
    n, f = 200, 6
    rng = np.random.Generator(np.random.PCG64(77))
    x = (rng.standard_normal((n, f)) * 4).astype(np.float32)
    ei = oracle.synthetic_edges(n, 3000, seed=7)
    ei = np.concatenate([ei, np.stack([np.arange(10, dtype=np.int32)] * 2)], axis=1)   # explicit self-loops too
    wq, wk = oracle.glorot_uniform(rng, f, 4) * 3, oracle.glorot_uniform(rng, f, 4) * 3
    bq = bk = np.zeros(4, np.float32)
    wv = oracle.glorot_uniform(rng, f, 8)
    got = tfg.nn.gat(x, ei, wq, bq, None, wk, bk, None, wv, None, None, num_heads=2).cpu().numpy()
    ref = oracle.gat(x, ei, wq, bq, None, wk, bk, None, wv, None, None, num_heads=2)
    q, k = oracle.matmul(x, wq), oracle.matmul(x, wk)
    smax = float(np.abs(q).max() * np.abs(k).max() * 2 / np.sqrt(2.0))
    assert smax > 20
    assert_parity(got, ref, tol=1e-5 * max(1.0, smax / 4), what="GAT large scores")


@pytest.mark.parametrize("cls,fn", [("MeanGraphSage", "mean_graph_sage"), ("SumGraphSage", "sum_graph_sage")])
@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("weighted", [True, False])
def test_sage_mean_sum(tfg, oracle, cls, fn, concat, weighted):
    x, ei, w, rng = _graph(oracle, 500, 6000, 100, seed=8)
    layer = getattr(tfg.layers, cls)(64, concat=concat, normalize=True)
    layer._maybe_build([x])
    ku = 32 if concat else 64
    ws, wn = oracle.glorot_uniform(rng, 100, ku), oracle.glorot_uniform(rng, 100, ku)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    layer.set_weights(self_kernel=ws, neighbor_kernel=wn, bias=b)
    inputs = [x, ei, w] if weighted else [x, ei]
    got = layer(inputs).cpu().numpy()
    ref = getattr(oracle, fn)(x, ei, w if weighted else None, ws, wn, b, "relu", concat=concat, normalize=True)
    assert_parity(got, ref, what="{} concat={} weighted={}".format(cls, concat, weighted))


@pytest.mark.parametrize("cls,fn,names", [
    ("MeanPoolGraphSage", "mean_pool_graph_sage", ("neighbor_mlp_kernel", "neighbor_mlp_bias", "neighbor_kernel")),
    ("MaxPoolGraphSage", "max_pool_graph_sage", ("neighbor_mlp_kernel", "neighbor_mlp_bias", "neighbor_kernel"))])
@pytest.mark.parametrize("concat", [True, False])
def test_sage_pool(tfg, oracle, cls, fn, names, concat):
    n = 300
    x, ei, w, rng = _graph(oracle, n, 6000, 40, seed=9)
    # every node gets an in-edge: an isolated node keeps float lowest() into the next GEMM (overflow by design)
    ring = np.stack([np.arange(n, dtype=np.int32), np.roll(np.arange(n, dtype=np.int32), 1)])
    ei = np.concatenate([ei, ring], axis=1)
    w = np.concatenate([w, np.ones(n, np.float32)])
    layer = getattr(tfg.layers, cls)(32, concat=concat)
    layer._maybe_build([x])
    ku = 16 if concat else 32
    ws = oracle.glorot_uniform(rng, 40, ku)
    wm = oracle.glorot_uniform(rng, 40, 4 * ku)
    bm = (rng.standard_normal(4 * ku) * 0.1).astype(np.float32)
    wn = oracle.glorot_uniform(rng, 4 * ku, ku)
    b = (rng.standard_normal(32) * 0.1).astype(np.float32)
    layer.set_weights(self_kernel=ws, neighbor_mlp_kernel=wm, neighbor_mlp_bias=bm, neighbor_kernel=wn, bias=b)
    got = layer([x, ei, w]).cpu().numpy()
    ref = getattr(oracle, fn)(x, ei, w, ws, wm, wn, bm, b, "relu", concat=concat)
    assert_parity(got, ref, what="{} concat={}".format(cls, concat))
    with pytest.raises(TypeError):
        layer([x, ei])      # edge_weight=None fails in the reference too (graph_sage.py:197/260)


@pytest.mark.parametrize("cache", [None, {}, {"x": 1}])
def test_gcn_graph_sage_quirks(tfg, oracle, cache):
    x, ei, w, rng = _graph(oracle, 300, 3000, 20, seed=10)
    k = oracle.glorot_uniform(rng, 20, 12)
    b = (rng.standard_normal(12) * 0.1).astype(np.float32)
    import copy
    user_cache = copy.copy(cache)       # the product adds its own "tfgx_*" bookkeeping entries to the dict it is given;
    got = tfg.nn.gcn_graph_sage(x, ei, w, k, b, tfg.relu, normalize=True, cache=cache).cpu().numpy()
    ref = oracle.gcn_graph_sage(x, ei, w, k, b, "relu", normalize=True, cache=user_cache)
    assert_parity(got, ref, what="gcn_graph_sage cache={}".format(user_cache))
    # those entries must not flip the reference's cache-lands-in-renorm quirk (graph_sage.py:142) on later calls
    got2 = tfg.nn.gcn_graph_sage(x, ei, w, k, b, tfg.relu, normalize=True, cache=cache).cpu().numpy()
    assert np.array_equal(got, got2)


def test_sparse_matrix_surface(tfg, oracle):
    x, ei, w, rng = _graph(oracle, 200, 2000, 10, seed=14)
    A = tfg.SparseMatrix(ei, w, [200, 200])
    assert_parity((A @ x).cpu().numpy(), oracle.spmm(ei, w, (200, 200), x), what="A @ x")
    assert_parity(A.segment_sum(axis=-1).cpu().numpy(), oracle.unsorted_segment_sum(w, ei[0], 200), what="rowsum")
    assert_parity(A.segment_sum(axis=0).cpu().numpy(), oracle.unsorted_segment_sum(w, ei[1], 200), what="colsum")
    ei2, w2 = oracle.add_self_loop_edge(ei, 200, w, 2.0)
    assert_parity((A.add_diag(2.0) @ x).cpu().numpy(), oracle.spmm(ei2, w2, (200, 200), x), what="add_diag")
    sm = A.segment_softmax(axis=-1)
    assert_parity(sm.value.cpu().numpy(), oracle.segment_softmax(w, ei[0], 200), what="segment_softmax")
    assert_parity((A.transpose() @ x).cpu().numpy(), oracle.spmm(ei[::-1], w, (200, 200), x), what="transpose")


def test_map_reduce_gnn_layer(tfg, oracle):
    x, ei, w, rng = _graph(oracle, 150, 1200, 6, seed=15)

    class MyGNN(tfg.layers.MapReduceGNN):
        def map(self, repeated_x, neighbor_x, edge_weight=None):
            return tfg.nn.gcn_mapper(repeated_x, neighbor_x, edge_weight)

        def reduce(self, neighbor_msg, node_index, num_nodes=None):
            return tfg.nn.sum_reducer(neighbor_msg, node_index, num_nodes)

        def update(self, x, reduced_neighbor_msg):
            return tfg.nn.sum_updater(x, reduced_neighbor_msg)

    got = MyGNN()([x, ei, w]).cpu().numpy()
    ref = oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer, oracle.sum_updater)
    assert_parity(got, ref, what="MapReduceGNN")


def test_gat_hub_rows_chunked_merge(tfg, oracle):
    """A destination with 6000 in-edges (and one with 700) takes the chunk + merge path; results match the oracle."""
    n, f = 1500, 10
    rng = np.random.Generator(np.random.PCG64(91))
    x = rng.standard_normal((n, f), dtype=np.float32)
    hub1 = np.stack([np.full(6000, 7, np.int32), rng.integers(0, n, 6000, dtype=np.int32)])
    hub2 = np.stack([np.full(700, 1200, np.int32), rng.integers(0, n, 700, dtype=np.int32)])
    ei = np.concatenate([hub1, oracle.synthetic_edges(n, 6000, seed=5), hub2], axis=1).astype(np.int32)
    from tf_geometric_amd.plan import CsrPlan
    plan = CsrPlan.build(ei, n, n)
    assert plan.hub_info() is not None and int(plan.hub_info()[0].shape[0]) >= 2
    for heads, att, units in [(4, 8, 16), (1, 3, 5), (2, 32, 8)]:
        wq, wk = oracle.glorot_uniform(rng, f, att), oracle.glorot_uniform(rng, f, att)
        bq = (rng.standard_normal(att) * 0.2).astype(np.float32)
        wv = oracle.glorot_uniform(rng, f, units)
        b = (rng.standard_normal(units) * 0.1).astype(np.float32)
        got = tfg.nn.gat(x, ei, wq, bq, tfg.relu, wk, bq, tfg.relu, wv, b, tfg.relu, num_heads=heads,
                         cache={"tfgx_csr_plan": plan}).cpu().numpy()
        ref = oracle.gat(x, ei, wq, bq, "relu", wk, bq, "relu", wv, b, "relu", num_heads=heads)
        assert_parity(got, ref, tol=2e-5, what="GAT hub H={}".format(heads))


def test_hip_graph_replay_of_two_layer_gcn(tfg, oracle):
    """A 2-layer GCN forward over a cached plan is captured into a hipGraph and replayed on new inputs."""
    import torch
    x, ei, w, rng = _graph(oracle, 3000, 40000, 32, seed=33)
    l0, l1 = tfg.layers.GCN(16, activation=tfg.relu), tfg.layers.GCN(7)
    cache = {}

    def model(xx):
        return l1([l0([xx, ei, w], cache=cache), ei, w], cache=cache)

    eager = model(tfg._lib.as_f32(x)).clone()
    cap = tfg.CapturedForward(model, x)
    assert torch.equal(cap(x), eager)
    x2 = (x * 0.5 + 1.0).astype(np.float32)
    got = cap(x2).clone()
    assert torch.equal(got, model(tfg._lib.as_f32(x2)))
    ref = oracle.gcn(oracle.gcn(x2, ei, w, l0.kernel.cpu().numpy(), l0.bias.cpu().numpy(), "relu"), ei, w,
                     l1.kernel.cpu().numpy(), l1.bias.cpu().numpy())
    assert_parity(got.cpu().numpy(), ref, what="captured 2-layer GCN")


def test_gemm_column_limited_activation(tfg, oracle):
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(5))
    a = rng.standard_normal((700, 40), dtype=np.float32)
    b = oracle.glorot_uniform(rng, 40, 80)
    bias = (rng.standard_normal(80) * 0.1).astype(np.float32)
    got = gemm_bias_act(a, b, bias=bias, act=1, act_cols=16).cpu().numpy()
    ref = oracle.matmul(a, b) + bias
    ref[:, :16] = np.maximum(ref[:, :16], 0)
    assert_parity(got, ref, what="act_cols")
    assert (got[:, 16:] < 0).any()


@pytest.mark.parametrize("m,k,n", [(33000, 100, 256), (40001, 36, 100), (50000, 128, 200), (32768, 100, 65), (70000, 20, 129),
                                   (33001, 60, 96), (40000, 32, 256), (35000, 128, 256), (33333, 256, 128),
                                   (34000, 44, 224), (36000, 64, 160), (32800, 92, 192), (33000, 256, 256), (33000, 200, 384)])
def test_gemm_streaming_kernel(tfg, oracle, m, k, n):
    """Tall-skinny shapes (M >= 32768, 64 < N <= 256, 32 <= K, K % 4 == 0, B within LDS) take the persistent
    row-streaming kernel: full steps only (K % 32 == 0), every tail length class, ragged M and N."""
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(m + k))
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = oracle.glorot_uniform(rng, k, n)
    bias = (rng.standard_normal(n) * 0.1).astype(np.float32)
    got = gemm_bias_act(a, b, bias=bias, act=1).cpu().numpy()
    ref = np.maximum(oracle.matmul(a, b) + bias, 0)
    assert_parity(got, ref, what="stream gemm {}x{}x{}".format(m, k, n))
    got2 = gemm_bias_act(a, b, act=1, act_cols=n // 2).cpu().numpy()
    ref2 = oracle.matmul(a, b)
    ref2[:, :n // 2] = np.maximum(ref2[:, :n // 2], 0)
    assert_parity(got2, ref2, what="stream gemm act_cols")


@pytest.mark.parametrize("m,k,n", [(33000, 1433, 16), (40007, 602, 8), (32768, 301, 7), (50001, 1024, 16), (33000, 257, 1),
                                   (36000, 1900, 12), (32769, 272, 16), (40000, 301, 40), (33000, 1433, 32), (35000, 602, 41),
                                   (33333, 67, 48), (34000, 777, 17)])
def test_gemm_long_k_narrow_output_kernel(tfg, oracle, m, k, n):
    """Narrow outputs (N <= 48: one to three 16-column MFMA tiles) of rows that are not 16-byte aligned (Cora-width 1433 -> 16,
    GAT's 602 -> 8 / 41): the persistent gemm_skinny_kernel — B resident in LDS, A streamed straight into the 16 x 16 x 4 MFMA
    layout, two-level sum.
    Every K tail class (K % 16 in 0 / 1 / 9 / 10 / 12 / 13), ragged M, bias / ReLU / column-limited activation; also a view
    of A with a leading dimension (rows even further from any alignment)."""
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(m + k))
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = oracle.glorot_uniform(rng, k, n)
    bias = (rng.standard_normal(n) * 0.1).astype(np.float32)
    got = gemm_bias_act(a, b, bias=bias, act=1).cpu().numpy()
    ref = np.maximum(oracle.matmul(a, b) + bias, 0)
    assert_parity(got, ref, what="long-K gemm {}x{}x{}".format(m, k, n))
    if n > 1:
        got2 = gemm_bias_act(a, b, act=1, act_cols=n // 2).cpu().numpy()
        ref2 = oracle.matmul(a, b)
        ref2[:, :n // 2] = np.maximum(ref2[:, :n // 2], 0)
        assert_parity(got2, ref2, what="long-K gemm act_cols")
    import torch
    wide = torch.zeros((m, k + 3), device="cuda")
    wide[:, :k] = torch.from_numpy(a).cuda()
    got3 = gemm_bias_act(wide[:, :k], b).cpu().numpy()
    assert_parity(got3, oracle.matmul(a, b), what="long-K gemm on a strided view")


def test_gcn_sparse_node_features(tfg, oracle):
    """gcn.py:269-270: sparse x (bag-of-words rows) -> x @ W as a segment-sum over the kernel rows; SparseMatrix and
    torch sparse COO inputs, forward parity with the dense path and the kernel gradient."""
    import torch
    rng = np.random.Generator(np.random.PCG64(21))
    n, f, u = 300, 50, 7
    dense = (rng.random((n, f)) < 0.08) * rng.standard_normal((n, f))
    dense = dense.astype(np.float32)
    dense[5] = 0                                                    # a node without features
    ei = oracle.synthetic_edges(n, 3000, seed=2)
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    k = oracle.glorot_uniform(rng, f, u)
    b = (rng.standard_normal(u) * 0.1).astype(np.float32)
    ref = oracle.gcn(dense, ei, w, k, b, "relu")
    r, c = np.nonzero(dense)
    xs = tfg.SparseMatrix(np.stack([r, c]).astype(np.int32), dense[r, c], [n, f])
    adj = tfg.SparseMatrix(ei, w, [n, n])
    got = tfg.nn.gcn(xs, adj, k, b, activation=tfg.relu)
    assert_parity(got.cpu().numpy(), ref, what="gcn sparse x (SparseMatrix)")
    xt = torch.sparse_coo_tensor(np.stack([r, c]), dense[r, c], (n, f)).cuda()
    got2 = tfg.nn.gcn(xt, adj, k, b, activation=tfg.relu)
    assert_parity(got2.cpu().numpy(), ref, what="gcn sparse x (torch COO)")
    kt = torch.tensor(k, device="cuda", requires_grad=True)
    out = tfg.nn.gcn(xs, adj, kt, torch.tensor(b, device="cuda"), activation=tfg.relu)
    out.square().sum().backward()
    kd = torch.tensor(k, device="cuda", requires_grad=True)
    out_d = tfg.nn.gcn(torch.tensor(dense, device="cuda"), adj, kd, torch.tensor(b, device="cuda"), activation=tfg.relu)
    out_d.square().sum().backward()
    assert_parity(kt.grad.cpu().numpy(), kd.grad.cpu().numpy(), tol=1e-4, what="d/dkernel through sparse x")
    layer = tfg.layers.GCN(u, activation=tfg.relu)
    layer._maybe_build([xs])
    layer.set_weights(kernel=k, bias=b)
    assert_parity(layer([xs, ei, w]).cpu().numpy(), ref, what="layers.GCN with sparse x")


def test_gcn_cache_builders_and_old_api(tfg, oracle):
    """gcn_build_cache_for_graph / gcn_build_cache_by_adj / gcn_norm_edge / gcn_cache_normed_edge (gcn.py:133-218) fill
    the same cache key the layer reads, and the layer then reuses it (same output, no second normalisation)."""
    class G(object):
        pass
    rng = np.random.Generator(np.random.PCG64(31))
    n = 200
    g_ = G()
    g_.edge_index = oracle.synthetic_edges(n, 1500, seed=5)
    g_.edge_weight = rng.uniform(0.5, 1.5, g_.edge_index.shape[1]).astype(np.float32)
    g_.x = rng.standard_normal((n, 6), dtype=np.float32)
    g_.num_nodes, g_.cache = n, {}
    cache = tfg.nn.gcn_build_cache_for_graph(g_)
    key = tfg.nn.conv.gcn.compute_cache_key("both", True, True, True, False) if hasattr(tfg.nn, "conv") else None
    assert cache is g_.cache and len(cache) >= 1
    idx, val = tfg.nn.gcn_norm_edge(g_.edge_index, n, g_.edge_weight, cache=g_.cache)
    oi, ov = oracle.gcn_norm_adj(g_.edge_index, g_.edge_weight, n)
    dense = np.zeros((n, n))
    np.add.at(dense, (idx.cpu().numpy()[0], idx.cpu().numpy()[1]), val.cpu().numpy())
    ref = np.zeros((n, n))
    np.add.at(ref, (oi[0], oi[1]), ov)
    assert_parity(dense.astype(np.float32), ref.astype(np.float32), what="gcn_norm_edge")
    before = dict(g_.cache)
    tfg.nn.gcn_cache_normed_edge(g_)                         # already cached: nothing changes
    assert all(g_.cache[k] is before[k] for k in before)
    tfg.nn.gcn_cache_normed_edge(g_, override=True)          # recomputed
    k0 = [k for k in before if k.startswith("gcn_normed_adj")][0]
    assert g_.cache[k0] is not None and g_.cache[k0] is not before[k0]
    layer = tfg.layers.GCN(4)
    kernel = oracle.glorot_uniform(rng, 6, 4)
    layer._maybe_build([g_.x])
    layer.set_weights(kernel=kernel, bias=np.zeros(4, np.float32))
    assert_parity(layer([g_.x, g_.edge_index, g_.edge_weight], cache=g_.cache).cpu().numpy(),
                  oracle.gcn(g_.x, g_.edge_index, g_.edge_weight, kernel), what="GCN on a prebuilt cache")


def test_static_aggregation_memo_is_opt_in_exact_and_invalidated(tfg, oracle):
    """prepare_static_features(..., cache_aggregation=True): layer 0's A_hat @ x (GCN, aggregation-first route) and the
    neighbour mean of x (GraphSAGE) are computed once and reused while x and the edge weights do not change — same bits as
    the plain path, in inference and in training (weight gradients equal), recomputed after a torch-visible update of
    x, never used for another tensor, gone after release_static_features."""
    import torch
    from tf_geometric_amd import plan as P
    rng = np.random.Generator(np.random.PCG64(31))
    n, f = 500, 20
    ei = oracle.synthetic_edges(n, 6000, seed=31)
    x = torch.tensor(rng.standard_normal((n, f)).astype(np.float32), device="cuda")
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    gcn = tfg.layers.GCN(48, activation=tfg.relu)            # 20 < 48: aggregation first
    sage = tfg.layers.MeanGraphSage(64, activation=tfg.relu)   # ku = 32 >= 20: reduce at the input width
    plain_cache = {}
    fused_gcn, fused_sage = gcn([x, ei, w], cache=plain_cache), sage([x, ei, w], cache=plain_cache)
    # "same bits" is a statement about the memo against the TWO-LAUNCH plain path (aggregate, then GEMM): the one-launch
    # fused form of the plain path (plan.aggregate_gemm) multiplies in another k order and agrees to fp32 rounding
    P.FUSE_AGGREGATE_GEMM = False
    o_gcn, o_sage = gcn([x, ei, w], cache=plain_cache), sage([x, ei, w], cache=plain_cache)
    assert_parity(fused_gcn.cpu().numpy(), o_gcn.cpu().numpy(), what="fused vs two-launch GCN layer")
    assert_parity(fused_sage.cpu().numpy(), o_sage.cpu().numpy(), what="fused vs two-launch SAGE layer")
    cache = {}
    tfg.prepare_static_features(x, ei, cache)                 # layout only: no memo
    gcn([x, ei, w], cache=cache)
    assert "tfgx_static_aggregated" not in cache
    tfg.prepare_static_features(x, ei, cache, cache_aggregation=True)
    P.STATIC_STATS["agg_hits"] = 0
    for _ in range(3):
        assert torch.equal(gcn([x, ei, w], cache=cache), o_gcn) and torch.equal(sage([x, ei, w], cache=cache), o_sage)
    assert P.STATIC_STATS["agg_hits"] == 4                    # first call of each layer computes, the rest hit
    other = x.clone()
    assert torch.equal(gcn([other, ei, w], cache=cache), o_gcn) and P.STATIC_STATS["agg_hits"] == 4
    # training: the memo feeds the differentiable GEMM; gradients equal the plain route's
    def grads(c):
        for layer in (gcn, sage):
            layer.trainable(True)
            for p_ in layer.parameters():
                p_.grad = None
        (gcn([x, ei, w], cache=c).sum() + sage([x, ei, w], cache=c).sum()).backward()
        return [p_.grad.clone() for p_ in gcn.parameters() + sage.parameters()]
    for a, b in zip(grads(cache), grads(plain_cache)):
        assert torch.equal(a, b)
    assert P.STATIC_STATS["agg_hits"] == 6
    x.mul_(1.5)                                               # torch-visible update: recomputed, not stale
    with torch.no_grad():
        fresh = gcn([x, ei, w], cache=cache)
        assert torch.equal(fresh, gcn([x.clone(), ei, w], cache={}))
    tfg.release_static_features(cache)
    assert "tfgx_static_aggregated" not in cache
    P.FUSE_AGGREGATE_GEMM = True


@pytest.mark.parametrize("n,e,f,units", [(5000, 60000, 100, 256), (777, 9000, 64, 40), (3000, 30000, 128, 128),
                                          (130, 900, 36, 7), (64, 300, 4, 1), (10000, 150000, 100, 128), (1000, 0, 8, 16),
                                          (6000, 70000, 128, 256), (1500, 20000, 128, 200), (900, 8000, 104, 129),
                                          (2100, 30000, 124, 256)])
@pytest.mark.parametrize("mode", ["gcn", "mean", "sum_unweighted"])
def test_fused_aggregate_gemm_equals_two_launches(tfg, oracle, n, e, f, units, mode):
    """tfgx_aggregate_gemm_f32 (aggregate -> LDS -> MFMA in one launch) vs tfgx_segment_reduce_f32 + tfgx_gemm_bias_act_f32
    and vs the float64 oracle: weighted sum with the implicit self-loop (GCN), mean, unweighted sum; tiles that are not
    full, empty rows, F not a multiple of the lane-group width, one output column."""
    import torch
    from tf_geometric_amd import plan as P
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(n + f))
    ei = oracle.synthetic_edges(n, e, seed=f) if e else np.zeros((2, 0), np.int32)
    if e:
        ei = ei[:, ei[0] % 11 != 3]                                       # rows without in-edges
    x = rng.standard_normal((n, f), dtype=np.float32)
    k = oracle.glorot_uniform(rng, f, units)
    b = (rng.standard_normal(units) * 0.1).astype(np.float32)
    plan = P.CsrPlan.build(L.as_i32(ei), n, n)
    xd, kd, bd = L.as_f32(x), L.as_f32(k), L.as_f32(b)
    w = rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32)
    if mode == "gcn":
        w_csr, sc, op = plan.edge_attr_to_csr(w), torch.rand(n, device="cuda") + 0.25, L.SUM
    elif mode == "mean":
        w_csr, sc, op = plan.edge_attr_to_csr(w), None, L.MEAN
    else:
        w_csr, sc, op = None, None, L.SUM
    assert L.require_gpu().tfgx_aggregate_gemm_fits(f, units) == 1
    fused = P.aggregate_gemm(plan, xd, op, kd, w_csr=w_csr, self_coef=sc, bias=bd, act=L.ACT_RELU)
    assert fused is not None and tuple(fused.shape) == (n, units)
    agg = P.segment_reduce(plan, xd, op, w_csr=w_csr, self_coef=sc)
    two = P.gemm_bias_act(agg, kd, bias=bd, act=L.ACT_RELU)
    # reference side: the oracle's aggregate (float64 accumulation over the caller's edge list, stored as float32 as the
    # reference stores it), float64 projection
    if ei.shape[1]:
        red = oracle.mean_reducer if mode == "mean" else oracle.sum_reducer
        agg_ref = oracle.aggregate_neighbors(x, ei, None if mode == "sum_unweighted" else w,
                                             oracle.identity_mapper if mode == "sum_unweighted" else oracle.gcn_mapper, red,
                                             oracle.identity_updater, num_nodes=n).astype(np.float64)
    else:
        agg_ref = np.zeros((n, f))
    if sc is not None:
        agg_ref = agg_ref + sc.double().cpu().numpy()[:, None] * x.astype(np.float64)
    assert_parity(agg.cpu().numpy(), agg_ref, what="two-launch aggregate vs oracle")
    ref = np.maximum(agg_ref @ k.astype(np.float64) + b, 0)
    assert_parity(fused.cpu().numpy(), ref, what="fused aggregate->gemm vs the oracle's aggregate projected in float64")
    assert_parity(fused.cpu().numpy(), two.cpu().numpy(), what="fused vs two launches")
    # into a column block of a wider output (GraphSAGE's concat halves) and without bias / activation
    wide = torch.full((n, units + 5), 7.0, device="cuda")
    P.aggregate_gemm(plan, xd, op, kd, w_csr=w_csr, self_coef=sc, out=wide[:, 5:])
    assert_parity(wide[:, 5:].cpu().numpy(), agg_ref @ k.astype(np.float64), what="fused into a column block")
    assert bool((wide[:, :5] == 7.0).all())
    # training forward: the aggregate itself as a side output — the bits tfgx_segment_reduce_f32 writes — same projection
    side = torch.full((n, f), float("nan"), device="cuda")
    again = P.aggregate_gemm(plan, xd, op, kd, w_csr=w_csr, self_coef=sc, bias=bd, act=L.ACT_RELU, agg_out=side)
    assert torch.equal(side, agg) and torch.equal(again, fused)
    # the static feature layout as the source (main rows of whole lines + node tails, then the per-edge tail stream): the
    # same FMA chain per element, the same projection order -> bit-identical to the launch on the dense table
    if f > 32 and f % 32 and ei.shape[1]:
        rows = P.SplitRows.from_dense(xd)
        assert torch.equal(P.aggregate_gemm(plan, rows, op, kd, w_csr=w_csr, self_coef=sc, bias=bd, act=L.ACT_RELU), fused)
        rows.with_edge_tail(plan)
        side2 = torch.empty_like(side)
        tail = P.aggregate_gemm(plan, rows, op, kd, w_csr=w_csr, self_coef=sc, bias=bd, act=L.ACT_RELU, agg_out=side2)
        assert torch.equal(tail, fused) and torch.equal(side2, agg)


def test_fused_aggregate_gemm_declines_what_it_cannot_take(tfg, oracle):
    import torch
    from tf_geometric_amd import plan as P
    L = tfg._lib
    lib = L.require_gpu()
    assert lib.tfgx_aggregate_gemm_fits(128, 256) == 1 and lib.tfgx_aggregate_gemm_fits(100, 256) == 1   # (B partly resident)
    assert lib.tfgx_aggregate_gemm_fits(102, 16) == 0 and lib.tfgx_aggregate_gemm_fits(132, 16) == 0
    assert lib.tfgx_aggregate_gemm_fits(100, 257) == 0
    ei = oracle.synthetic_edges(500, 4000, seed=1)
    plan = P.CsrPlan.build(L.as_i32(ei), 500, 500)
    x = torch.randn(500, 128, device="cuda")
    assert P.aggregate_gemm(plan, x, L.SUM, torch.randn(128, 300, device="cuda")) is None      # more than 256 output columns
    assert P.aggregate_gemm(plan, x[:, :126], L.SUM, torch.randn(126, 16, device="cuda")) is None   # F % 4 != 0
    assert P.aggregate_gemm(plan, x, L.MAX, torch.randn(128, 16, device="cuda")) is None       # max is not linear


@pytest.mark.parametrize("mode", ["gcn", "mean"])
@pytest.mark.parametrize("f,units,thr", [(100, 256, 64), (64, 40, 16), (36, 128, 128), (100, 41, 64)])
def test_fused_aggregate_gemm_on_a_graph_with_hub_rows(tfg, oracle, mode, f, units, thr):
    """Power-law graphs: rows longer than the plan's hub threshold are cut into chunks, reduced chunk by chunk by a launch of
    the ordinary kernel, and folded in chunk order by the row's lane group inside the fused launch — same rows as the two
    launches (whose hub path folds the same partials in the same order) and as the float64 product.  The plan is walked in
    degree order and n = 4000 leaves a ragged last tile: units % 4 == 0 takes the 16-byte transposed stores with the row
    ids from LDS, units = 41 the per-column stores."""
    import torch
    from tf_geometric_amd import plan as P
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(f + thr))
    n = 4000
    ei = oracle.synthetic_edges(n, 40000, seed=thr)
    hubs = np.stack([rng.integers(0, 25, size=30000, dtype=np.int32),            # 25 destinations with ~1200 extra in-edges
                     rng.integers(0, n, size=30000, dtype=np.int32)])
    ei = np.concatenate([ei, hubs], axis=1)
    x = rng.standard_normal((n, f), dtype=np.float32)
    k, b = oracle.glorot_uniform(rng, f, units), (rng.standard_normal(units) * 0.1).astype(np.float32)
    old = (P.HUB_THRESHOLD, P.HUB_CHUNK)
    P.HUB_THRESHOLD, P.HUB_CHUNK = thr, max(8, thr // 2)
    try:
        plan = P.CsrPlan.build(L.as_i32(ei), n, n)
        hub = plan.hub_info()
        assert hub is not None and int(hub[0].shape[0]) >= 25
        xd, kd, bd = L.as_f32(x), L.as_f32(k), L.as_f32(b)
        # (weights ~ 1 / hub degree: a 1200-term row stays O(1), so the 1e-5 band tests the kernel, not fp32 at magnitude 100)
        w_csr = plan.edge_attr_to_csr(rng.uniform(0.5, 1.5, ei.shape[1]).astype(np.float32) / (40.0 if mode == "gcn" else 1.0))
        sc = L.as_f32(rng.uniform(0.25, 1.25, n).astype(np.float32)) if mode == "gcn" else None
        op = L.SUM if mode == "gcn" else L.MEAN
        fused = P.aggregate_gemm(plan, xd, op, kd, w_csr=w_csr, self_coef=sc, bias=bd, act=L.ACT_RELU)
        assert fused is not None
        agg = P.segment_reduce(plan, xd, op, w_csr=w_csr, self_coef=sc)
        two = P.gemm_bias_act(agg, kd, bias=bd, act=L.ACT_RELU)
    finally:
        P.HUB_THRESHOLD, P.HUB_CHUNK = old
    ref = np.maximum(agg.double().cpu().numpy() @ k.astype(np.float64) + b, 0)
    assert_parity(fused.cpu().numpy(), ref, what="fused with hub rows vs float64 of the same aggregate")
    assert_parity(fused.cpu().numpy(), two.cpu().numpy(), what="fused with hub rows vs two launches")


@pytest.mark.parametrize("m,k,n", [(300001, 100, 256), (280000, 256, 256), (262144, 128, 96), (270000, 36, 40)])
def test_gemm_dynamic_tile_order_changes_no_bit(tfg, oracle, m, k, n):
    """Tall products (M >= 2^18) given a workspace run the row kernel with CLAIMED tiles (per-pool device counters in the
    workspace) instead of the fixed tile -> wave map of the workspace-less entry point: a tile's arithmetic does not depend on
    the wave that runs it, so the two launches agree in every bit; and both sit in the oracle's band."""
    import torch
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.plan import gemm_bias_act
    lib = L.require_gpu()
    assert lib.tfgx_gemm_workspace_bytes(m, k, n) > 0
    rng = np.random.Generator(np.random.PCG64(m + n))
    a = torch.as_tensor(rng.standard_normal((m, k), dtype=np.float32), device="cuda")
    b = torch.as_tensor(oracle.glorot_uniform(rng, k, n), device="cuda")
    bias = torch.as_tensor((rng.standard_normal(n) * 0.1).astype(np.float32), device="cuda")
    dyn = gemm_bias_act(a, b, bias=bias, act=1)                                     # workspace lent: claimed tiles
    fixed = torch.empty_like(dyn)
    L.check(lib.tfgx_gemm_bias_act_f32(L.ptr(a), k, L.ptr(b), n, L.ptr(bias), 1, L.ptr(fixed), n, m, k, n, L.stream_ptr()),
            "tfgx_gemm_bias_act_f32")
    assert torch.equal(dyn, fixed)
    rows = rng.choice(m, size=4000, replace=False)
    ref = np.maximum(oracle.matmul(a[rows].cpu().numpy(), b.cpu().numpy()) + bias.cpu().numpy(), 0)
    assert_parity(dyn[rows].cpu().numpy(), ref, what="dynamic-order gemm {}x{}x{}".format(m, k, n))
    again = gemm_bias_act(a, b, bias=bias, act=1)                                   # and run to run
    assert torch.equal(dyn, again)


def test_layer_losses_collect_the_regularisers_as_keras_does(tfg, oracle):
    """kernel_regularizer goes to every glorot-initialised matrix, bias_regularizer to every zero-initialised vector
    (layers/conv/gcn.py:26-30, gat.py:64-83, graph_sage.py:55-61); `layer.losses` holds one term per regularised weight,
    follows set_weights, and is differentiable on a trainable layer (the demos' hand-written L2 term, demo_gcn.py:60-66)."""
    import torch
    x, ei, w, rng = _graph(oracle, 300, 2000, 20, seed=8)
    l2 = lambda t: 5e-4 * (t * t).sum() / 2            # noqa: E731
    l1 = lambda t: 1e-3 * t.abs().sum()                # noqa: E731

    gcn = tfg.layers.GCN(8, kernel_regularizer=l2, bias_regularizer=l1)
    assert gcn.losses == []                            # nothing built yet
    gcn._maybe_build([x])
    kernel = oracle.glorot_uniform(rng, 20, 8)
    bias = (rng.standard_normal(8) * 0.1).astype(np.float32)
    gcn.set_weights(kernel=kernel, bias=bias)
    got = sorted(float(t) for t in gcn.losses)
    want = sorted([5e-4 * float((kernel.astype(np.float64) ** 2).sum()) / 2, 1e-3 * float(np.abs(bias).sum())])
    assert np.allclose(got, want, rtol=1e-5)
    assert tfg.layers.GCN(8)([x, ei, w]) is not None and tfg.layers.GCN(8).losses == []    # no regulariser, no term

    gat = tfg.layers.GAT(8, num_heads=2, kernel_regularizer=l2)
    gat._maybe_build([x])
    assert len(gat.losses) == 3                        # query_kernel, key_kernel, kernel; the biases take none here
    sage = tfg.layers.MaxPoolGraphSage(8, kernel_regularizer=l2, bias_regularizer=l1)
    sage._maybe_build([x])
    assert len(sage.losses) == len([t for t in sage.weights.values() if t is not None])

    gcn.trainable(True)
    out = gcn([x, ei, w])
    loss = (out * out).mean() + sum(gcn.losses)
    loss.backward()
    gk = gcn.kernel.grad.clone()
    gcn.kernel.grad = None
    gcn.bias.grad = None
    (gcn([x, ei, w]) ** 2).mean().backward()
    extra = (gk - gcn.kernel.grad).cpu().numpy()       # d/dW of 5e-4 * |W|^2 / 2 = 5e-4 * W
    assert np.allclose(extra, 5e-4 * kernel, atol=1e-7)
    assert torch.isfinite(gcn.bias.grad).all()


@pytest.mark.parametrize("m,k,n,act_cols", [(5000, 602, 160, 160), (4100, 301, 144, 130), (4096, 257, 288, 100), (6000, 1433, 192, 192)])
def test_gemm_short_column_remainder_runs_as_a_second_product(tfg, oracle, m, k, n, act_cols):
    """N = q * 128 + r with r <= 64 on a long, unaligned K (round 6): the first q * 128 columns and the remainder are two
    launches (no half-empty last tile column); bias, ReLU and the column-limited activation land in the right columns."""
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(m + n))
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = oracle.glorot_uniform(rng, k, n)
    bias = (rng.standard_normal(n) * 0.1).astype(np.float32)
    got = gemm_bias_act(a, b, bias=bias, act=1, act_cols=act_cols).cpu().numpy()
    ref = oracle.matmul(a, b) + bias
    ref[:, :act_cols] = np.maximum(ref[:, :act_cols], 0)
    assert_parity(got, ref, what="gemm {}x{}x{} (act_cols {})".format(m, k, n, act_cols))
    if act_cols < n:
        assert (got[:, act_cols:] < 0).any()
