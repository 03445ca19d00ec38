# coding=utf-8
"""2-layer GCN on a Cora-SHAPED synthetic graph — the MI355X counterpart of the reference's demo/demo_gcn.py
(BASELINE.json configs[0]; real Cora needs a download, there is no network here).

Same model (GCN(16, relu) -> GCN(num_classes), dropout 0.5, Adam 1e-2, L2 5e-4 on kernels), same layer call
signature `[x, edge_index, edge_weight], cache=graph_cache`, same closing "mean forward time" measurement
(demo/demo_gcn.py:99-105).  tf.GradientTape -> torch.autograd over the kernels' own backward (tf_geometric_amd.autograd).

    python examples/demo_gcn.py [--steps 200] [--hipgraph]
    python examples/demo_gcn.py --shape products [--steps 6]

--shape products: the same loop on the ogbn-products-shaped synthetic graph (N = 2.4 M, E = 123 M, F = 100, hidden 256),
full batch, no dropout on the input features — the case the reference's loop (demo/demo_gcn.py:68-77) presents to layer 0:
the SAME feature tensor in every step.  No API of this package is called beyond the reference's own layer signature; the
log shows layer 0's forward time per step and when the static feature layout was promoted (plan.AUTO_STATIC_LAYOUT: built
on the second sighting of the tensor, used from then on).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_geometric_amd as tfg   # noqa: E402


def cora_shaped(seed=0, n=2708, e=10556, f=1433, classes=7):
    """Planted-partition graph with bag-of-words-like sparse features that carry a weak class signal."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y = rng.integers(0, classes, size=n)
    half = e // 2
    a = rng.integers(0, n, size=half)
    same = rng.random(half) < 0.8                                   # 80% of edges stay inside a class (homophily)
    order = np.argsort(y, kind="stable")                            # nodes of a class are contiguous in `order`
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    near = order[np.clip(pos[a] + rng.integers(-40, 41, half), 0, n - 1)]
    b = np.where(same, near, rng.integers(0, n, size=half))
    keep = a != b
    a, b = a[keep], b[keep]
    edge_index = np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)
    x = (rng.random((n, f)) < 0.012).astype(np.float32)             # ~17 active words per node, like Cora
    for c in range(classes):                                        # class-specific vocabulary block
        cols = slice(c * 60, c * 60 + 60)
        x[y == c, cols] = (rng.random((int((y == c).sum()), 60)) < 0.12).astype(np.float32)
    x = x / np.maximum(x.sum(1, keepdims=True), 1.0)
    idx = rng.permutation(n)
    return x, edge_index, y.astype(np.int64), idx[:140], idx[140:640], idx[1708:]


class GCNModel(object):
    def __init__(self, num_classes, drop_rate=0.5):
        self.gcn0 = tfg.layers.GCN(16, activation=tfg.relu)
        self.gcn1 = tfg.layers.GCN(num_classes)
        self.drop_rate = drop_rate

    def __call__(self, inputs, training=False, cache=None):
        x, edge_index, edge_weight = inputs
        h = torch.nn.functional.dropout(x, self.drop_rate, training)
        h = self.gcn0([h, edge_index, edge_weight], cache=cache)
        h = torch.nn.functional.dropout(h, self.drop_rate, training)
        return self.gcn1([h, edge_index, edge_weight], cache=cache)

    def parameters(self):
        return self.gcn0.parameters() + self.gcn1.parameters()


def main(steps=200, forward_iters=1000, quiet=False, hipgraph=False):
    x_np, edge_index, y_np, train_index, valid_index, test_index = cora_shaped()
    num_classes = int(y_np.max()) + 1
    x = tfg._lib.as_f32(x_np)
    y = torch.as_tensor(y_np, device=x.device)
    edge_weight = np.ones(edge_index.shape[1], dtype=np.float32)    # Graph default (data/graph.py:53-56)
    cache = {}                                                       # graph.cache
    model = GCNModel(num_classes)
    model([x, edge_index, edge_weight], cache=cache)                 # builds weights + plan + normalised adjacency
    model.gcn0.trainable(True)
    model.gcn1.trainable(True)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=hipgraph)
    tr, te = torch.as_tensor(train_index, device=x.device), torch.as_tensor(test_index, device=x.device)

    def evaluate():
        with torch.no_grad():
            logits = model([x, edge_index, edge_weight], cache=cache)
        return float((logits[te].argmax(-1) == y[te]).float().mean())

    y_tr = y[tr]

    def compute_loss():
        logits = model([x, edge_index, edge_weight], training=True, cache=cache)
        loss = torch.nn.functional.cross_entropy(logits[tr], y_tr)
        return loss + 5e-4 * sum(0.5 * (p ** 2).sum() for p in (model.gcn0.kernel, model.gcn1.kernel))

    def eager_step():
        optimizer.zero_grad()
        loss = compute_loss()
        loss.backward()
        optimizer.step()
        return loss

    # --hipgraph: the whole step (zero-grad, dropout, forward, loss, backward, Adam) replayed from ONE hipGraph — what
    # tf.function buys the reference's training forward (demo/demo_gcn.py:64-66, "10X faster" :107-109); at this size a
    # step is ~50 launches of a few microseconds each, i.e. pure launch cost
    train_step = tfg.CapturedTrainStep(compute_loss, optimizer) if hipgraph else eager_step
    acc = evaluate()
    t_train, timed = None, 0
    for step in range(1, steps + 1):
        if step == min(4, steps):        # the first steps build lazily-built plan metadata (transposed plan ...): not timed
            torch.cuda.synchronize()
            t_train, timed = time.time(), steps - step + 1
        loss = train_step()
        if step % 20 == 0:
            acc = evaluate()
            if not quiet:
                print("step = {}\tloss = {:.4f}\taccuracy = {:.4f}".format(step, float(loss.detach()), acc))
    torch.cuda.synchronize()
    if not quiet and steps:
        print("mean training step time ({}): {:.6f} seconds".format("hipGraph replay" if hipgraph else "eager",
                                                                    (time.time() - t_train) / timed))
    mean_forward = None
    if forward_iters:
        with torch.no_grad():
            torch.cuda.synchronize()
            start = time.time()
            for _ in range(forward_iters):
                model([x, edge_index, edge_weight], cache=cache)
            torch.cuda.synchronize()
            mean_forward = (time.time() - start) / forward_iters
        if not quiet:
            print("mean forward time: {:.6f} seconds".format(mean_forward))
    return acc, mean_forward


def main_products(steps=6, quiet=False):
    """Full-batch training of GCN(256, relu) -> GCN(47) at products shape through the reference's layer signature only."""
    from tf_geometric_amd import synthetic, plan as P
    n, e, f = synthetic.WORKLOADS["products"]
    classes = 47
    edge_index = tfg._lib.as_i32(synthetic.synthetic_edges(n, e, seed=0))
    x = tfg._lib.as_f32(synthetic.synthetic_features(n, f, seed=1))
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    y = torch.randint(0, classes, (n,), generator=g, device="cuda")
    tr = torch.arange(0, n, 10, device="cuda")
    cache = {}
    gcn0, gcn1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(classes)
    with torch.no_grad():
        gcn1([gcn0([x, edge_index], cache=cache), edge_index], cache=cache)      # builds weights + plan + normalised adjacency
    tfg.release_static_features(cache)                                           # (the build call above counts as no sighting)
    cache.pop("tfgx_static_seen", None)
    gcn0.trainable(True)
    gcn1.trainable(True)
    optimizer = torch.optim.Adam(gcn0.parameters() + gcn1.parameters(), lr=1e-2)
    log = []
    for step in range(1, steps + 1):
        optimizer.zero_grad()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        h = gcn0([x, edge_index], cache=cache)                                    # layer 0: the static input features
        e1.record()
        logits = gcn1([h, edge_index], cache=cache)
        loss = torch.nn.functional.cross_entropy(logits[tr], y[tr])
        loss.backward()
        optimizer.step()
        e2.record()
        torch.cuda.synchronize()
        rec = dict(step=step, loss=float(loss.detach()), layer0_forward_ms=e0.elapsed_time(e1), step_ms=e0.elapsed_time(e2),
                   static_layout="edge_tail" if cache.get("tfgx_static_rows", (None, None))[1] is not None else "none",
                   auto_promotions=P.STATIC_STATS.get("auto_promotions", 0))
        log.append(rec)
        if not quiet:
            print("step = {step}\tloss = {loss:.4f}\tlayer-0 forward = {layer0_forward_ms:.2f} ms\tstep = {step_ms:.2f} ms\t"
                  "static layout = {static_layout} (automatic promotions so far: {auto_promotions})".format(**rec))
    return log


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--hipgraph", action="store_true", help="replay the whole training step from one hipGraph")
    ap.add_argument("--shape", default="cora", choices=["cora", "products"])
    args = ap.parse_args()
    if args.shape == "products":
        main_products(steps=args.steps or 6)
    else:
        main(steps=args.steps or 200, hipgraph=args.hipgraph)
