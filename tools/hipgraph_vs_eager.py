# coding=utf-8
"""hipGraph replay vs eager launches of the 2-layer GCN forward (F -> 256 -> 40), phase by phase, so that a rocprofv3
kernel trace of this script can be cut into [eager] and [replay] segments (the phases are separated by 0.3 s of idle).

    python tools/hipgraph_vs_eager.py [products|arxiv] [--static]
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/graph_trace -- python tools/hipgraph_vs_eager.py products
    python tools/trace_gaps.py gpurun_out/graph_trace         # per-phase kernel time, gaps between kernels, wall
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg            # noqa: E402
from tf_geometric_amd import synthetic    # noqa: E402

which = next((a for a in sys.argv[1:] if not a.startswith("--")), "products")
static = "--static" in sys.argv
steps = 10
n, e, f = synthetic.WORKLOADS[which]
ei = tfg._lib.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.randn(n, f, generator=g, device="cuda")
cache = {}
g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)


def two_layer():
    return g1([g0([x, ei], cache=cache), ei], cache=cache)


two_layer()
if static:
    tfg.prepare_static_features(x, ei, cache)
for _ in range(3):
    two_layer()
cap = tfg.CapturedForward(two_layer)
for _ in range(3):
    cap.graph.replay()
torch.cuda.synchronize()


def phase(fn):
    time.sleep(0.3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, t_issue * 1e3 / steps


res = {"workload": which, "static_layout": static, "steps": steps}
for rep in range(2):                      # eager, replay, eager, replay: order effects show up as a difference
    res["eager_ms_{}".format(rep)], res["eager_issue_ms_{}".format(rep)] = phase(two_layer)
    res["replay_ms_{}".format(rep)], res["replay_issue_ms_{}".format(rep)] = phase(lambda: cap.graph.replay())
print(json.dumps(res))
