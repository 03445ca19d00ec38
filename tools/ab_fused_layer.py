# coding=utf-8
"""A/B of the fused aggregate->GEMM launch (tfgx_aggregate_gemm_f32) against the two launches it replaces, alternating in
one process: GCN layer 100 -> 256 (+ bias, ReLU) and the mean-SAGE layer (units 256, concat) at products shape; the
aggregation alone and the GEMM alone beside them.

    python tools/ab_fused_layer.py [products|arxiv] [uniform|rmat] [F]  > gpurun_out/r04/ab_fused_layer.json
F overrides the workload's feature width (128: BASELINE configs C2 / C5 — half of the 128 x 256 kernel is read from L2).
Also times ONE TRAINING step of the GCN layer (forward through the fused launch with the aggregate as a side output,
backward) against the un-fused training route.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                          # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P     # noqa: E402
import bench                                            # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "products"
graph = sys.argv[2] if len(sys.argv) > 2 else "uniform"          # "rmat": power-law in-degrees (hub rows, chunked)
n, e, f = synthetic.WORKLOADS[which]
if len(sys.argv) > 3:
    f = int(sys.argv[3])
ei = synthetic.rmat_edges(n, e, 7, torch.device("cuda")) if graph == "rmat" else L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.randn(n, f, generator=g, device="cuda")
cache = {}
P.AUTO_STATIC_LAYOUT = False      # the first block times the launches on x AS IT IS (no promotion to the static layout on the
                                  # layers' second call); the *_static keys are measured after the explicit opt-in below
gcn = tfg.layers.GCN(256, activation=tfg.relu)
sage = tfg.layers.MeanGraphSage(256)
w1 = torch.ones(int(ei.shape[1]), device="cuda")
gcn([x, ei], cache=cache)
sage([x, ei, w1], cache=cache)
from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj   # noqa: E402
normed = gcn_norm_adj(cache["tfgx_gcn_adj"], cache=cache)
k = torch.randn(f, 256, device="cuda") * 0.1
agg_out = torch.empty(n, f, device="cuda")


def set_fuse(v):
    P.FUSE_AGGREGATE_GEMM = v


fns = {
    "gcn_layer_fused": lambda: (set_fuse(True), gcn([x, ei], cache=cache)),
    "gcn_layer_two_launches": lambda: (set_fuse(False), gcn([x, ei], cache=cache)),
    "mean_sage_layer_fused": lambda: (set_fuse(True), sage([x, ei, w1], cache=cache)),
    "mean_sage_layer_two_launches": lambda: (set_fuse(False), sage([x, ei, w1], cache=cache)),
    "aggregation_alone": lambda: P.segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=agg_out),
    "gemm_alone": lambda: P.gemm_bias_act(agg_out, k),
}
gt = tfg.layers.GCN(256, activation=tfg.relu)
gt._maybe_build([x])
gt.trainable(True)


def train_step(fuse):
    set_fuse(fuse)
    for p_ in gt.parameters():
        p_.grad = None
    gt([x, ei], cache=cache).sum().backward()


fns["gcn_layer_fwd_bwd_fused_forward"] = lambda: train_step(True)
fns["gcn_layer_fwd_bwd_two_launch_forward"] = lambda: train_step(False)
times = {name: [] for name in fns}
for rnd in range(4):
    for name, fn in fns.items():
        times[name].append(bench._time(fn, steps=10, warmup=3 if rnd == 0 else 1))
set_fuse(True)
med = {name: sorted(v)[len(v) // 2] for name, v in times.items()}
# the same layers with x in the static feature layout (what a layer's second call promotes to): fused launch on the split /
# edge-tail source rows (round 4) against segment-reduce on that layout + GEMM
info = tfg.prepare_static_features(x, ei, cache)
if info["layout"] == "edge_tail":
    fns_s = {"gcn_layer_fused_static": fns["gcn_layer_fused"], "gcn_layer_two_launches_static": fns["gcn_layer_two_launches"],
             "mean_sage_layer_fused_static": fns["mean_sage_layer_fused"],
             "mean_sage_layer_two_launches_static": fns["mean_sage_layer_two_launches"],
             "gcn_layer_fwd_bwd_fused_forward_static": fns["gcn_layer_fwd_bwd_fused_forward"],
             "gcn_layer_fwd_bwd_two_launch_forward_static": fns["gcn_layer_fwd_bwd_two_launch_forward"]}
    ts = {name: [] for name in fns_s}
    for rnd in range(4):
        for name, fn in fns_s.items():
            ts[name].append(bench._time(fn, steps=10, warmup=3 if rnd == 0 else 1))
    set_fuse(True)
    med.update({name: sorted(v)[len(v) // 2] for name, v in ts.items()})
    times.update(ts)
tfg.release_static_features(cache)
a, b = gcn([x, ei], cache=cache), None
set_fuse(False)
b = gcn([x, ei], cache=cache)
set_fuse(True)
print(json.dumps({"shape": which, "graph": graph, "N": n, "E": int(ei.shape[1]), "F": f, "ms_median_of_4_alternating_rounds": med,
                  "max_abs_diff_fused_vs_two_launches": float((a - b).abs().max()), "all_rounds_ms": times}))
