# coding=utf-8
"""2-layer MeanGraphSage with per-layer neighbour sampling (k = 25, 10) on PPI-shaped synthetic graphs — counterpart of
the reference's demo/demo_graph_sage.py: RandomNeighborSampler per graph, MeanGraphSage(256, relu, concat) x 2, a
dropout + dense head, multi-label sigmoid loss + 1e-5 L2 on the kernels, Adam 1e-2, micro-F1 (demo lines 14-47, 50-70).
The sampler draws a new edge list every forward (one kernel launch, tfgx_sample_neighbors), so every step builds a
fresh CSR plan: this demo exercises sampler -> plan -> aggregation -> backward end to end.

    python examples/demo_graph_sage.py [--epochs 10]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_geometric_amd as tfg   # noqa: E402
from tf_geometric_amd.utils import RandomNeighborSampler   # noqa: E402

NUM_SAMPLED = [25, 10]


def ppi_shaped(num_graphs, seed, n=2200, avg_deg=28, f=50, classes=121, mix=None):
    """Graphs whose labels depend on a node's own features AND on its neighbourhood mean (so aggregation helps)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if mix is None:
        mix = np.random.Generator(np.random.PCG64(1234))
    w_self, w_nb = mix.standard_normal((f, classes)), mix.standard_normal((f, classes))
    graphs = []
    for _ in range(num_graphs):
        e = n * avg_deg // 2
        a, b = rng.integers(0, n, e), rng.integers(0, n, e)
        keep = a != b
        a, b = a[keep], b[keep]
        ei = np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)
        x = rng.standard_normal((n, f)).astype(np.float32)
        deg = np.maximum(np.bincount(ei[0], minlength=n), 1)[:, None]
        nb = np.zeros((n, f))
        np.add.at(nb, ei[0], x[ei[1]])
        y = ((x @ w_self + 3.0 * (nb / deg) @ w_nb) > 0).astype(np.float32)
        graphs.append(dict(x=x, edge_index=ei, y=y))
    return graphs


def micro_f1(y_true, logits):
    pred = logits > 0
    tp = float((pred & (y_true > 0)).sum())
    return 2 * tp / max(float(pred.sum()) + float((y_true > 0).sum()), 1.0)


def main(epochs=10, quiet=False, num_train=4):
    torch.manual_seed(0)
    train, test = ppi_shaped(num_train, seed=1), ppi_shaped(1, seed=2)
    for g in train + test:                                        # traverse all graphs (demo :15-18)
        g["sampler"] = RandomNeighborSampler(tfg._lib.as_i32(g["edge_index"]))   # device tensors in -> device tensors out
        g["xt"], g["yt"] = tfg._lib.as_f32(g["x"]), tfg._lib.as_f32(g["y"])
    num_classes = train[0]["y"].shape[1]
    sages = [tfg.layers.MeanGraphSage(units=256, activation=tfg.relu, concat=True),
             tfg.layers.MeanGraphSage(units=256, activation=tfg.relu, concat=True)]
    for s in sages:
        s.trainable(True)
    fc = torch.nn.Sequential(torch.nn.Dropout(0.3), torch.nn.Linear(256, num_classes)).cuda()
    step_seed = [0]

    def forward(g, training=False):
        h = g["xt"]
        for sage, k in zip(sages, NUM_SAMPLED):
            step_seed[0] += 1
            ei, ew = g["sampler"].sample(k=k, seed=step_seed[0])  # a fresh sample per layer and per call (demo :55)
            h = sage([h, ei, ew], training=training)
        fc.train(training)
        return fc(h)

    with torch.no_grad():
        forward(train[0])                                          # lazy weight build
    params = [p for s in sages for p in s.parameters()] + list(fc.parameters())
    kernels = [p for s in sages for n_, p in s.weights.items() if "kernel" in n_] + [fc[1].weight]
    opt = torch.optim.Adam(params, lr=1e-2)
    f1, loss = 0.0, None
    for epoch in range(epochs):
        for g in train:
            opt.zero_grad()
            logits = forward(g, training=True)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, g["yt"])
            loss = loss + 1e-5 * sum(0.5 * (p ** 2).sum() for p in kernels)
            loss.backward()
            opt.step()
        with torch.no_grad():
            f1 = micro_f1(test[0]["yt"], forward(test[0]))
        if not quiet:
            print("epoch = {}\tloss = {:.4f}\ttest_f1_micro = {:.4f}".format(epoch, float(loss.detach()), f1))
    return f1, float(loss.detach())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10)
    main(epochs=ap.parse_args().epochs)
