# coding=utf-8
"""What do the per-class passes of the sharded step cost against ONE pass over the same edges?  One GPU, the products-shaped
graph, ShardedGraph in self-halo test mode: only the first n/W rows are resident sources, the other (W-1)/W of the source
rows arrive through the halo exchange (from the rank itself) in R rounds — the pass structure of rank 0 of W at EVERY row
(n_own = N here, so divide the times by W for a per-rank figure).  Times: the single launch, the R + 1 passes without any
exchange (exchange=False on a filled table), and the full step with the exchange through the product transport.

    python tools/shard_pass_probe.py [W ...]      -> one JSON line per W
"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from tf_geometric_amd import synthetic, _lib as L                     # noqa: E402
from tf_geometric_amd.plan import CsrPlan, segment_reduce             # noqa: E402
from tf_geometric_amd.dist.sharded import ShardedGraph                # noqa: E402
import bench                                                          # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
plan = CsrPlan.build(ei, n, n)
w = torch.rand(plan.num_edges, device="cuda")
out1 = torch.empty(n, f, device="cuda")
single = bench._time(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out1), steps=10, warmup=3)
for W in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    for R in (4, 2, 1):
        sg = ShardedGraph.from_global(ei, n, edge_weight=None, rounds=R, self_halo_rows=n // W)
        table = sg.alloc_table(f)
        sg.own_rows(table).copy_(x)
        h = sg.exchange_start(table)
        sg.exchange_finish(h)
        out = torch.empty(sg.n_own, f, device="cuda")
        passes = bench._time(lambda: sg.aggregate(table, L.SUM, w=None, out=out, exchange=False), steps=10, warmup=3)
        step = bench._time(lambda: sg.aggregate(table, L.SUM, w=None, out=out), steps=10, warmup=3)
        per_class = [bench._time(lambda k=k: sg.aggregate(table, L.SUM, w=None, out=out, exchange=False, classes=[k]),
                                 steps=10, warmup=2) for k in range(sg.n_class)]
        print(json.dumps({"W": W, "rounds": R, "single_launch_weighted_ms": round(single, 3), "passes_ms": round(passes, 3),
                          "passes_over_single": round(passes / single, 3), "step_with_self_exchange_ms": round(step, 3),
                          "per_class_ms": [round(v, 3) for v in per_class], "halo_rows": int(sg.n_halo),
                          "transport": sg.transport.name}), flush=True)
        del sg, table, out
dist.destroy_process_group()
