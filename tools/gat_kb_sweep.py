# coding=utf-8
"""Reddit-shaped GAT attention (H = 8; A = 8 and 64, W = 64): the number of source blocks swept around the policy's choice.
One JSON line per (A, KB)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import synthetic, _lib as L, plan as P              # noqa: E402
from tf_geometric_amd.nn.conv import gat as G                              # noqa: E402

n, e, f = synthetic.WORKLOADS["reddit"]
dev = torch.device("cuda")
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
cache = {}
plan = P.CsrPlan.from_cache(ei, n, n, cache)
torch.manual_seed(3)


def timeit(fn, steps=8, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for A in (8, 64):
    Q, K, V = torch.randn(n, A, device=dev), torch.randn(n, A, device=dev), torch.randn(n, 64, device=dev)
    G.SOURCE_BLOCKS = None
    policy = G.source_block_count(plan, A, 64)
    for kb in (1, 4, 6, 8, 10, 12, 14, 16, 20, 24):
        G.SOURCE_BLOCKS = kb
        print(json.dumps({"A": A, "KB": kb, "policy_KB": policy,
                          "attention_ms": round(timeit(lambda: G.gat_attention(plan, Q, K, V, 8)), 4)}), flush=True)
G.SOURCE_BLOCKS = None
