# coding=utf-8
"""Scan the steps around the hot path (edge preprocessing, sampler, readouts, sparse features, plan build, normalisation)
at large sizes; prints ms per call and the implied rate so that a pathological fallback stands out."""
import gc
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L


def t(fn, k=3, warm=1):
    for _ in range(warm):
        fn()
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    gc.enable()
    return (time.perf_counter() - t0) * 1e3 / k


def out(name, ms, items, unit):
    print(json.dumps({"what": name, "ms": round(ms, 3), "rate": "{:.2f} M {}/s".format(items / ms / 1e3, unit)}), flush=True)


n, e = 1000000, 50000000
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
E = int(ei.shape[1])
w = torch.rand(E, device="cuda") + 0.5
U = tfg.utils
out("CsrPlan.build", t(lambda: tfg.CsrPlan.build(ei, n, n)), E, "edges")
out("gcn_norm_adj", t(lambda: tfg.nn.gcn_norm_adj(tfg.SparseMatrix(ei, w, [n, n]))), E, "edges")
out("remove_self_loop_edge", t(lambda: U.remove_self_loop_edge(ei, w)), E, "edges")
out("add_self_loop_edge", t(lambda: U.add_self_loop_edge(ei, n, w)), E, "edges")
half = ei[:, : E // 2].contiguous()
out("convert_edge_to_directed", t(lambda: U.convert_edge_to_directed(half, [w[: E // 2]])), E // 2, "edges")
out("merge_duplicated_edge(sum)", t(lambda: U.merge_duplicated_edge(ei, [w], ["sum"])), E, "edges")
out("merge_duplicated_edge(max)", t(lambda: U.merge_duplicated_edge(ei, [w], ["max"])), E, "edges")
out("get_laplacian(sym)", t(lambda: U.get_laplacian(ei, n, w, "sym")), E, "edges")
s = U.RandomNeighborSampler(ei, w)
out("RandomNeighborSampler(k=10)", t(lambda: s.sample(k=10, seed=1)), n * 10, "samples")
out("RandomNeighborSampler(ratio=0.5)", t(lambda: s.sample(ratio=0.5, seed=1)), E // 2, "samples")
sub = torch.randperm(n, device="cuda")[: n // 10].to(torch.int32)
out("RandomNeighborSampler(k=25, 10% nodes)", t(lambda: s.sample(k=25, sampled_node_index=sub, seed=1)), (n // 10) * 25, "samples")
x = torch.randn(n, 64, device="cuda")
gi = torch.sort(torch.randint(0, 50000, (n,), device="cuda"))[0].to(torch.int32)
for name in ("mean_pool", "sum_pool", "max_pool", "min_pool"):
    fn = getattr(tfg.nn, name)
    out(name + " (1M nodes -> 50k graphs, F=64)", t(lambda: fn(x, gi, 50000)), n, "nodes")
score = torch.randn(n, device="cuda")
out("topk_pool(ratio=0.5)", t(lambda: tfg.nn.topk_pool(gi, score, ratio=0.5)), n, "nodes")
att = torch.randn(E, 8, device="cuda")
out("segment_softmax [E, 8]", t(lambda: tfg.nn.segment_softmax(att, ei[0], n)), E, "edges")
# sparse bag-of-words features (Cora-like density) through a GCN layer
nnz = n * 20
sp = tfg.SparseMatrix(torch.stack([torch.randint(0, n, (nnz,), device="cuda"), torch.randint(0, 1433, (nnz,), device="cuda")]).to(torch.int32),
                      torch.rand(nnz, device="cuda"), [n, 1433])
layer = tfg.layers.GCN(16, activation=tfg.relu)
cache = {}
out("GCN(16) on sparse x [1M, 1433], 20 nnz/row", t(lambda: layer([sp, ei, w], cache=cache)), E, "edges")
