# coding=utf-8
"""Plan build of ONE papers100M-shaped shard (BASELINE configs[4]), timed on one GPU.

A shard of the 8-way split owns 13.9 M destination rows and 200 M in-edges whose sources lie anywhere among 111 M nodes.
World size 1 cannot have remote peers, so the shard's plan work is reproduced with the constructors' self-halo test mode:
the first 13.9 M nodes are the resident own rows, every other referenced source becomes a halo row requested (through the
transport: RCCL self send / receive) from the rank itself.  What is timed is exactly what a rank of the 8-GPU job runs:
in-degree histogram over 111 M nodes, split points, stable CSR sort of 200 M edges, halo mark / compact / remap over
111 M ids, request-list exchange, round-major layout, per-class edge partition — all on the device.

    python tools/shard_plan_build.py [--edges 200000000] > gpurun_out/r03/shard_plan_build.jsonl
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                  # noqa: E402
from tf_geometric_amd.dist.sharded import ShardedGraph           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=111000000)
ap.add_argument("--own", type=int, default=13875000)
ap.add_argument("--edges", type=int, default=200000000)
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
g = torch.Generator(device="cuda")
g.manual_seed(12)
dst = torch.randint(0, args.own, (args.edges,), generator=g, device="cuda", dtype=torch.int32)
src = torch.randint(0, args.nodes, (args.edges,), generator=g, device="cuda", dtype=torch.int32)
ei = torch.stack([dst, src])
del dst, src
w = torch.rand(args.edges, generator=g, device="cuda") + 0.5
torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    sg = ShardedGraph.from_partitioned(ei, args.nodes, edge_weight_part=w, rounds=args.rounds, self_halo_rows=args.own)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sg.build_gcn_norm()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"what": "papers100M-shaped shard plan build (self-halo mode, 1 x MI355X)", "rep": rep,
                      "nodes": args.nodes, "own_rows": args.own, "edges": args.edges, "rounds": sg.rounds,
                      "halo_rows": sg.n_halo, "rows_requested_through_transport": int(sum(sg.recv_counts)),
                      "transport": sg.transport.name, "plan_build_s": round(t1 - t0, 3),
                      "gcn_norm_s": round(t2 - t1, 3), "peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}),
          flush=True)
    del sg
