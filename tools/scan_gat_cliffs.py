# coding=utf-8
"""Scan GAT layer configurations for performance cliffs (a fast kernel's layout conditions not met -> generic
one-lane-per-(row, head) kernels).  Prints ns per edge for forward and forward + backward; outliers are cliffs."""
import gc
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

n, e, f = 100000, 10000000, 32
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
E = int(ei.shape[1])
x = torch.randn(n, f, device="cuda")
cache = {}


def t(fn, k=3):
    for _ in range(2):
        fn()
    gc.collect()
    gc.disable()      # a full cyclic collection of a torch process is a 40 ms host pause: keep it out of the timed steps
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    gc.enable()
    return a.elapsed_time(b) / k


for heads, att, units, split in [(8, 8, 64, True), (1, 1, 41, True), (1, 1, 7, True), (1, 3, 10, True), (2, 6, 10, True),
                                 (4, 16, 20, True), (3, 9, 9, True), (8, 256, 64, True), (4, 12, 64, True), (8, 64, 64, True),
                                 (2, 2, 82, True), (8, 8, 16, False), (4, 4, 10, False), (1, 8, 300, True), (6, 6, 6, True),
                                 (5, 20, 40, True), (8, 40, 64, True)]:
    layer = tfg.layers.GAT(units, num_heads=heads, attention_units=att, split_value_heads=split)
    fwd = t(lambda: layer([x, ei], cache=cache))
    layer.trainable(True)

    def step():
        for p_ in layer.parameters():
            p_.grad = None
        layer([x, ei], cache=cache).sum().backward()
    tr = t(step, k=2)
    print(json.dumps({"H": heads, "A": att, "U": units, "split": split, "fwd_ns_per_edge": round(fwd * 1e6 / E, 3),
                      "fwd_bwd_ns_per_edge": round(tr * 1e6 / E, 3)}), flush=True)
