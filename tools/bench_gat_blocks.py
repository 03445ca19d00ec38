# coding=utf-8
"""Reddit-shaped GAT layer (BASELINE configs[2]; demo/demo_gat.py:22 literal A = 8 and the heavy A = 64): one pass per
kernel against chained launches over source / destination blocks (nn/conv/gat.source_block_count), alternating in one
process: attention alone, layer forward, layer forward + backward.  One JSON line per variant."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                             # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P              # noqa: E402
from tf_geometric_amd.nn.conv import gat as G                              # noqa: E402

n, e, f = synthetic.WORKLOADS["reddit"]
dev = torch.device("cuda")
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
E = int(ei.shape[1])
torch.manual_seed(3)
x = torch.randn(n, f, device=dev)
cache = {}
plan = P.CsrPlan.from_cache(ei, n, n, cache)


def timeit(fn, steps=6, warmup=2):
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


for (H, A, U) in ((8, 8, 64), (8, 64, 64)):
    Q, K, V = torch.randn(n, A, device=dev), torch.randn(n, A, device=dev), torch.randn(n, U, device=dev)
    lay = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
    lay([x, ei], cache=cache)
    tl = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
    tl._maybe_build([x])
    tl.trainable(True)

    def fwd_bwd():
        for p_ in tl.parameters():
            p_.grad = None
        tl([x, ei], cache=cache).sum().backward()

    ref = None
    for rep in range(2):
        for setting in (1, None):          # 1 = always one pass; None = the policy (source blocks at this shape)
            G.SOURCE_BLOCKS = setting
            kb = G.source_block_count(plan, A, U)
            out = G.gat_attention(plan, Q, K, V, H)
            if ref is None:
                ref = out
            row = {"H": H, "A": A, "U": U, "rep": rep, "source_blocks": kb,
                   "attention_ms": timeit(lambda: G.gat_attention(plan, Q, K, V, H)),
                   "layer_forward_ms": timeit(lambda: lay([x, ei], cache=cache)),
                   "layer_fwd_bwd_ms": timeit(fwd_bwd, steps=4, warmup=2),
                   "max_abs_diff_vs_one_pass": float((out - ref).abs().max().item())}
            row["fwd_bwd_over_forward"] = row["layer_fwd_bwd_ms"] / row["layer_forward_ms"]
            print(json.dumps(row), flush=True)
    G.SOURCE_BLOCKS = None
