# coding=utf-8
"""2-layer GAT on the Cora-shaped synthetic graph — counterpart of the reference's demo/demo_gat.py: GAT(64, relu,
num_heads=8, attention_units=8, edge_drop_rate=0.6) -> GAT(num_classes, num_heads=1, attention_units=1,
edge_drop_rate=0.6), feature dropout 0.6, Adam 5e-3, L2 5e-4 on the kernels (demo/demo_gat.py:14-31,57-60).
Attention dropout runs inside the fused kernels in the forward and is regenerated in the backward kernels.

    python examples/demo_gat.py [--steps 200]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_geometric_amd as tfg   # noqa: E402
from demo_gcn import cora_shaped  # noqa: E402

DROP_RATE = 0.6


class GATModel(object):
    def __init__(self, num_classes):
        self.gat0 = tfg.layers.GAT(64, activation=tfg.relu, num_heads=8, attention_units=8, edge_drop_rate=DROP_RATE)
        self.gat1 = tfg.layers.GAT(num_classes, num_heads=1, attention_units=1, edge_drop_rate=DROP_RATE)
        for layer in (self.gat0, self.gat1):
            layer.trainable(True)                      # applies when the weights are created on the first call

    def __call__(self, inputs, training=False, cache=None):
        x, edge_index = inputs
        h = torch.nn.functional.dropout(x, DROP_RATE, training)
        h = self.gat0([h, edge_index], training=training, cache=cache)
        h = torch.nn.functional.dropout(h, DROP_RATE, training)
        return self.gat1([h, edge_index], training=training, cache=cache)

    def parameters(self):
        return self.gat0.parameters() + self.gat1.parameters()


def main(steps=200, quiet=False, seed=0):
    torch.manual_seed(seed)
    x_np, edge_index, y_np, train_index, _, test_index = cora_shaped()
    x = tfg._lib.as_f32(x_np)
    y = torch.as_tensor(y_np, device=x.device)
    cache = {}
    model = GATModel(int(y_np.max()) + 1)
    with torch.no_grad():
        model([x, edge_index], cache=cache)            # builds the weights and the CSR plan
    optimizer = torch.optim.Adam(model.parameters(), lr=5e-3)
    tr, te = torch.as_tensor(train_index, device=x.device), torch.as_tensor(test_index, device=x.device)

    def evaluate():
        with torch.no_grad():
            logits = model([x, edge_index], cache=cache)
        return float((logits[te].argmax(-1) == y[te]).float().mean())

    acc, loss = evaluate(), None
    for step in range(1, steps + 1):
        optimizer.zero_grad()
        logits = model([x, edge_index], training=True, cache=cache)
        loss = torch.nn.functional.cross_entropy(logits[tr], y[tr])
        kernels = [model.gat0.kernel, model.gat0.query_kernel, model.gat0.key_kernel,
                   model.gat1.kernel, model.gat1.query_kernel, model.gat1.key_kernel]
        loss = loss + 5e-4 * sum(0.5 * (p ** 2).sum() for p in kernels)
        loss.backward()
        optimizer.step()
        if step % 20 == 0:
            acc = evaluate()
            if not quiet:
                print("step = {}\tloss = {:.4f}\taccuracy = {:.4f}".format(step, float(loss.detach()), acc))
    return acc, float(loss.detach())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    main(steps=ap.parse_args().steps)
