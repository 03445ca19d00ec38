# coding=utf-8
"""Synthetic inputs of SURVEY.md §8d (shared by bench.py and the examples; numpy only)."""
import numpy as np


def synthetic_edges(num_nodes, num_edges, seed=0):
    """E/2 uniform pairs on [0,N)^2, self pairs dropped, both directions emitted as [all (a,b) | all (b,a)] — the
    layout tf_geometric's convert_edge_to_directed produces (utils/graph_utils.py:186-190). Duplicates are kept
    (they sum). int32 [2, ~E]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    half = num_edges // 2
    a = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    b = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    keep = a != b
    a, b = a[keep], b[keep]
    return np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)


def synthetic_features(num_nodes, num_features, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((num_nodes, num_features), dtype=np.float32)


def synthetic_edge_weight(num_edges, seed=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(0.5, 1.5, size=num_edges).astype(np.float32)


def glorot_uniform(fan_in, fan_out, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=(fan_in, fan_out)).astype(np.float32)


WORKLOADS = {
    # name: (nodes, edges, features)  — BASELINE.json configs
    "products": (2400000, 123000000, 100),   # ogbn-products-shaped: north-star target (segment-sum, 1 GPU)
    "arxiv": (170000, 1200000, 128),         # ogbn-arxiv-shaped (configs[1])
    "reddit": (233000, 114000000, 602),      # Reddit-shaped (configs[2]: 8-head GAT)
    "cora": (2708, 10556, 1433),             # Cora-shaped (configs[0])
    "tiny": (20000, 400000, 100),            # plumbing check
}
