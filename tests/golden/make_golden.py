# coding=utf-8
"""Generates tests/golden/hot_path_small.npz.

The reference cannot be imported here (it needs tensorflow + tf_sparse, both absent: SURVEY.md §8c), so these
vectors are produced by the float64-accumulated oracle restatement (oracle/tfg_oracle.py) — they pin the oracle and
the HIP path against regressions, and are the inputs/outputs to replay through the real tf_geometric once a box
with TensorFlow is available (tests/golden/replay_with_tensorflow.py documents how)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tfg_oracle as o   # noqa: E402

n, e, f = 64, 600, 12
ei = o.synthetic_edges(n, e, seed=42)
rng = np.random.Generator(np.random.PCG64(43))
x = rng.standard_normal((n, f), dtype=np.float32)
w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
d = dict(x=x, edge_index=ei, edge_weight=w)
d["gcn_kernel"], d["gcn_bias"] = o.glorot_uniform(rng, f, 8), (rng.standard_normal(8) * 0.1).astype(np.float32)
d["gcn_out"] = o.gcn(x, ei, w, d["gcn_kernel"], d["gcn_bias"], "relu")
d["max_out"] = o.aggregate_neighbors(x, ei, w, o.gcn_mapper, o.max_reducer, o.identity_updater)
d["sage_self"], d["sage_neigh"] = o.glorot_uniform(rng, f, 8), o.glorot_uniform(rng, f, 8)
d["sage_bias"] = (rng.standard_normal(16) * 0.1).astype(np.float32)
d["sage_out"] = o.mean_graph_sage(x, ei, w, d["sage_self"], d["sage_neigh"], d["sage_bias"], "relu", normalize=True)
d["gat_wq"], d["gat_wk"], d["gat_wv"] = o.glorot_uniform(rng, f, 8), o.glorot_uniform(rng, f, 8), o.glorot_uniform(rng, f, 16)
d["gat_bq"], d["gat_bk"] = (rng.standard_normal(8) * 0.1).astype(np.float32), (rng.standard_normal(8) * 0.1).astype(np.float32)
d["gat_b"] = (rng.standard_normal(16) * 0.1).astype(np.float32)
d["gat_out"] = o.gat(x, ei, d["gat_wq"], d["gat_bq"], "relu", d["gat_wk"], d["gat_bk"], "relu", d["gat_wv"], d["gat_b"],
                     "relu", num_heads=4)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hot_path_small.npz"), **d)
print("wrote hot_path_small.npz")
