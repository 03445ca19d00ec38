# coding=utf-8
"""Reddit-shaped GAT(64, heads 8, attention_units 8): the attention launches timed on the layer's OWN Q / K / V (strided views of
the fused [Q | K] projection, ReLU'd) beside contiguous random Q / K / V of the same shapes.  One JSON line per case."""
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
torch.manual_seed(3)
x = torch.randn(n, f, device=dev)
cache = {}
plan = P.CsrPlan.from_cache(ei, n, n, cache)


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


H, A, U = 8, 8, 64
lay = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
lay([x, ei], cache=cache)
Q, K, V = G._project_qkv(x, lay.query_kernel, lay.query_bias, tfg.relu, lay.key_kernel, lay.key_bias, tfg.relu, lay.kernel)
cases = {
    "layer views (Q, K strided in [Q | K], relu'd)": (Q, K, V),
    "layer values, contiguous copies": (Q.contiguous(), K.contiguous(), V.contiguous()),
    "randn, contiguous": (torch.randn(n, A, device=dev), torch.randn(n, A, device=dev), torch.randn(n, U, device=dev)),
    "randn in a [Q | K] table (strided views)": (lambda t: (t[:, :A], t[:, A:], torch.randn(n, U, device=dev)))(torch.randn(n, 2 * A, device=dev)),
}
for name, (q, k, v) in cases.items():
    row = {"case": name, "ldq": q.stride(0), "ldk": k.stride(0), "source_blocks": G.source_block_count(plan, A, U),
           "attention_ms": timeit(lambda: G.gat_attention(plan, q, k, v, H)),
           "attention_bias_relu_ms": timeit(lambda: G.gat_attention(plan, q, k, v, H, bias=lay.bias, act=L.ACT_RELU))}
    print(json.dumps(row), flush=True)
print(json.dumps({"layer_forward_ms": timeit(lambda: lay([x, ei], cache=cache)),
                  "projections_ms": timeit(lambda: G._project_qkv(x, lay.query_kernel, lay.query_bias, tfg.relu, lay.key_kernel,
                                                                   lay.key_bias, tfg.relu, lay.kernel))}), flush=True)
