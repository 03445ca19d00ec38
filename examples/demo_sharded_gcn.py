# coding=utf-8
"""2-layer GCN trained on ONE large graph sharded by destination-node range over the GPUs of a node — the MI355X counterpart
of the reference's demo/demo_distributed_gcn.py.

The reference's distributed demo replicates the whole graph on every GPU (tf.distribute.MirroredStrategy) and all-reduces
the gradients (demo_distributed_gcn.py:52-57,99).  Here the GRAPH is what is split: every rank owns a contiguous range of
destination rows (edge-balanced), gathers its neighbours' hidden rows through a halo exchange (RCCL all-to-all-v in rounds
on a second HIP stream, overlapped with the own-source pass; backward: the reverse exchange), and the replicated weights'
gradients are summed with one flat all-reduce — the one collective the reference performs.

    python examples/demo_sharded_gcn.py                       # one GPU (no exchange)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/demo_sharded_gcn.py
    TFGX_DEMO_SELF_HALO=1 python examples/demo_sharded_gcn.py  # one GPU, rows really travel through RCCL (test mode)
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_geometric_amd as tfg                                  # noqa: E402
from tf_geometric_amd.dist.sharded import ShardedGraph          # noqa: E402


def planted_graph(n, e, f, classes, seed=0):
    """Homophilous synthetic graph (80 % of the edges stay inside a class) with weakly informative features."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y = rng.integers(0, classes, size=n)
    half = e // 2
    a = rng.integers(0, n, size=half)
    same = rng.random(half) < 0.8
    order = np.argsort(y, kind="stable")
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    near = order[np.clip(pos[a] + rng.integers(-200, 201, half), 0, n - 1)]
    b = np.where(same, near, rng.integers(0, n, size=half))
    keep = a != b
    a, b = a[keep], b[keep]
    ei = np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    x[np.arange(n), y % f] += 1.5                               # one feature column leaks the class, noisily
    return x, ei, y.astype(np.int64)


def main(n=200000, e=4000000, f=64, hidden=64, classes=16, steps=60, quiet=False):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    self_halo = os.environ.get("TFGX_DEMO_SELF_HALO", "0") != "0"
    if (world > 1 or self_halo) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29751")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl")
    x_np, ei, y_np = planted_graph(n, e, f, classes)            # (every rank draws the same graph; a real job reads stripes:
    #                                                              ShardedGraph.from_partitioned)
    sg = ShardedGraph.from_global(ei, n, self_halo_rows=(n // 4 if self_halo else None))
    sg.build_gcn_norm()                                         # sharded gcn_norm_adj: D^-1/2 (A + I) D^-1/2
    lo, hi = sg.own_lo, sg.own_hi
    be = sg.backend
    x_own = be.f32(x_np[lo:hi])
    y_own = torch.as_tensor(y_np[lo:hi], device=x_own.device)
    rng = np.random.Generator(np.random.PCG64(1))
    is_train = torch.as_tensor(rng.random(n)[lo:hi] < 0.3, device=x_own.device)
    n_train = int(is_train.sum().item())
    tot = torch.tensor(float(n_train), device=x_own.device)
    if world > 1:
        dist.all_reduce(tot)
    g = torch.Generator(device="cpu")
    g.manual_seed(0)                                            # replicated weights: the same initial values on every rank
    k0 = (torch.rand(f, hidden, generator=g) * 2 - 1).mul_(np.sqrt(6.0 / (f + hidden))).to(x_own.device).requires_grad_(True)
    b0 = torch.zeros(hidden, device=x_own.device, requires_grad=True)
    k1 = (torch.rand(hidden, classes, generator=g) * 2 - 1).mul_(np.sqrt(6.0 / (hidden + classes))).to(x_own.device).requires_grad_(True)
    b1 = torch.zeros(classes, device=x_own.device, requires_grad=True)
    params = [k0, b0, k1, b1]
    opt = torch.optim.Adam(params, lr=1e-2)

    def forward():
        h = sg.gcn_trainable(x_own, k0, b0, torch.relu)          # [n_own, hidden]: own rows only, halo rows arrive by exchange
        return sg.gcn_trainable(h, k1, b1, None)

    acc = 0.0
    for step in range(1, steps + 1):
        opt.zero_grad(set_to_none=True)
        logits = forward()
        # sum over MY training rows / global count: the ranks' losses add up to the mean over all training nodes
        loss = torch.nn.functional.cross_entropy(logits[is_train], y_own[is_train], reduction="sum") / tot
        loss.backward()
        sg.all_reduce_gradients(params)                         # demo_distributed_gcn.py:99 (strategy.reduce of the tape's grads)
        opt.step()
        if step % 20 == 0 or step == steps:
            with torch.no_grad():
                pred = forward().argmax(-1)
                stat = torch.stack([(pred[~is_train] == y_own[~is_train]).float().sum(), (~is_train).float().sum(),
                                    loss.detach()])
            if world > 1:
                dist.all_reduce(stat)
            acc = float(stat[0] / stat[1])
            if rank == 0 and not quiet:
                print("step = {}\tloss = {:.4f}\ttest accuracy = {:.4f}\t(world {}, transport {}, {} halo rows on rank 0)".format(
                    step, float(stat[2]), acc, world, sg.transport.name, sg.n_halo))
    return acc


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--nodes", type=int, default=200000)
    ap.add_argument("--edges", type=int, default=4000000)
    a = ap.parse_args()
    main(n=a.nodes, e=a.edges, steps=a.steps)
    if dist.is_initialized():
        torch.cuda.synchronize()
        from tf_geometric_amd.dist.transport import close_transports
        close_transports()
        dist.barrier()
        dist.destroy_process_group()
