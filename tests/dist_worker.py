# coding=utf-8
"""Worker for the multi-process sharding tests (spawned by test_dist_gloo.py / test_gpu_dist.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("TFGX_DIST_DEBUG_CHECKS", "1")     # the early-halo-send consistency check of dist/sharded.py (synchronises)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_inputs(n=400, e=5000, f=12, seed=0, skew=False):
    from oracle import tfg_oracle as oracle
    ei = oracle.synthetic_edges(n, e, seed=seed)
    if skew:   # make the split points uneven: many edges into the first rows, some rows with no in-edges
        rng = np.random.Generator(np.random.PCG64(seed + 5))
        extra = np.stack([rng.integers(0, 20, size=e // 2, dtype=np.int32), rng.integers(0, n, size=e // 2, dtype=np.int32)])
        ei = np.concatenate([ei, extra], axis=1)
        ei = ei[:, ei[0] % 7 != 3]
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
    k = oracle.glorot_uniform(rng, f, 10)
    b = (rng.standard_normal(10) * 0.1).astype(np.float32)
    return ei, x, w, k, b


def gat_weights():
    from oracle import tfg_oracle as oracle
    rng = np.random.Generator(np.random.PCG64(77))
    return (oracle.glorot_uniform(rng, 12, 6), oracle.glorot_uniform(rng, 12, 6), oracle.glorot_uniform(rng, 12, 8),
            (rng.standard_normal(6) * 0.2).astype(np.float32))


GAT_HEADS = 2


def gat_train_weights():
    """(query_kernel, query_bias, key_kernel, key_bias, kernel, bias) of a GAT(8, heads 2, attention_units 6) layer."""
    from oracle import tfg_oracle as oracle
    rng = np.random.Generator(np.random.PCG64(78))
    return (oracle.glorot_uniform(rng, 12, 6), (rng.standard_normal(6) * 0.2).astype(np.float32),
            oracle.glorot_uniform(rng, 12, 6), (rng.standard_normal(6) * 0.2).astype(np.float32),
            oracle.glorot_uniform(rng, 12, 8), (rng.standard_normal(8) * 0.2).astype(np.float32))


def pool_weights():
    """(self_kernel, neighbor_mlp_kernel, neighbor_kernel, neighbor_mlp_bias, bias) of MaxPoolGraphSage(10, concat)."""
    from oracle import tfg_oracle as oracle
    rng = np.random.Generator(np.random.PCG64(79))
    # MLP width 32: wide enough for the tracked span-by-span max forward of the HIP backend (plan.can_track: F >= 32)
    return (oracle.glorot_uniform(rng, 12, 5), oracle.glorot_uniform(rng, 12, 32), oracle.glorot_uniform(rng, 32, 5),
            (rng.standard_normal(32) * 0.2).astype(np.float32), (rng.standard_normal(10) * 0.2).astype(np.float32))


def run_checks(rank, world, use_gpu, skew, results, rounds=None, partitioned=False, hub_threshold=None, self_halo=False,
               transport=None):
    """Build the shard, run sharded GCN / mean / max / sum and return this rank's rows (as numpy).  hub_threshold: force
    the chunked long-span paths (reduce passes AND the sharded GAT's part lists) on this small graph."""
    from tf_geometric_amd.dist.sharded import ShardedGraph
    import tf_geometric_amd.plan as P
    P.HUB_THRESHOLD, P.HUB_CHUNK = hub_threshold, (None if hub_threshold is None else max(2, hub_threshold // 2))
    ei, x, w, k, b = make_inputs(skew=skew)
    n = x.shape[0]
    if use_gpu:
        backend = None
    else:
        from cpu_backend import NumpyBackend
        backend = NumpyBackend()
    group = dist.group.WORLD if dist.is_initialized() else None

    # self_halo (test mode): only the first third of a rank's rows are resident sources, the rest come through the halo
    # exchange from the rank itself — a world-size-1 run then moves real rows through the transport
    kw = dict(group=group, backend=backend, rounds=rounds, transport=transport)

    def make(weights):
        sh = (n // world) // 3 if self_halo else None
        if not partitioned:
            return ShardedGraph.from_global(ei, n, edge_weight=weights, self_halo_rows=sh, **kw)
        # every rank holds a different, interleaved stripe of the edge list (destinations all over the graph)
        part = slice(rank, None, world)
        return ShardedGraph.from_partitioned(ei[:, part], n, edge_weight_part=None if weights is None else weights[part],
                                             self_halo_rows=sh, **kw)
    sg = make(w)
    assert sg.rounds == (0 if (world == 1 and not self_halo) else (rounds or 1))
    be = sg.backend
    x_own = be.f32(x[sg.own_lo:sg.own_hi])
    out = {"lo": sg.own_lo, "hi": sg.own_hi, "edges": sg.num_edges, "n_halo": sg.n_halo, "transport": sg.transport.name,
           "dense_send": int(sum(sg.dense_send)), "rows_packed": int(sg.send_idx_packed.shape[0]),
           "rows_sent": int(sum(sg.send_counts))}
    sg.build_gcn_norm()
    out["gcn"] = sg.gcn(x_own, be.f32(k), bias=be.f32(b), act=1).cpu().numpy()
    out["gcn_nokernel"] = sg.gcn(x_own, None).cpu().numpy()
    out["mean"] = sg.neighbor_reduce(x_own, 1).cpu().numpy()
    out["max"] = sg.neighbor_reduce(x_own, 2).cpu().numpy()
    out["sum_unweighted"] = sg.neighbor_reduce(x_own, 0, weighted=False).cpu().numpy()
    wq, wk, wv, bq = gat_weights()
    out["gat"] = sg.gat(x_own, be.f32(wq), be.f32(bq), 1, be.f32(wk), be.f32(bq), 1, be.f32(wv), bias=be.f32(b[:8]),
                        act=1, num_heads=2).cpu().numpy()
    rs = np.random.Generator(np.random.PCG64(99))
    f = x.shape[1]
    ks, kn = (rs.standard_normal((f, 5)) * 0.3).astype(np.float32), (rs.standard_normal((f, 5)) * 0.3).astype(np.float32)
    kmlp, bmlp = (rs.standard_normal((f, 8)) * 0.3).astype(np.float32), (rs.standard_normal(8) * 0.1).astype(np.float32)
    kn2, b10 = (rs.standard_normal((8, 5)) * 0.3).astype(np.float32), (rs.standard_normal(10) * 0.1).astype(np.float32)
    out["sage_mean"] = sg.graph_sage(x_own, be.f32(ks), be.f32(kn), bias=be.f32(b10), act=1).cpu().numpy()
    out["sage_sum_add"] = sg.graph_sage(x_own, be.f32(ks), be.f32(kn), bias=be.f32(b10[:5]), concat=False,
                                        op=0).cpu().numpy()
    out["sage_max_pool"] = sg.pool_graph_sage(x_own, be.f32(ks), be.f32(kmlp), be.f32(kn2), be.f32(bmlp),
                                              bias=be.f32(b10), act=1).cpu().numpy()
    out["sage_mean_pool"] = sg.pool_graph_sage(x_own, be.f32(ks), be.f32(kmlp), be.f32(kn2), be.f32(bmlp),
                                               bias=be.f32(b10), act=1, op=1).cpu().numpy()       # projects before it gathers
    out["sage_mean_pool_add"] = sg.pool_graph_sage(x_own, be.f32(ks), be.f32(kmlp), be.f32(kn2), be.f32(bmlp),
                                                   bias=be.f32(b10[:5]), act=1, op=1, concat=False).cpu().numpy()
    sg2 = make(None)
    sg2.build_gcn_norm(norm="left", improved=True)
    out["gcn_left_improved_unweighted"] = sg2.gcn(x_own, be.f32(k)).cpu().numpy()
    sg3 = make(w)
    sg3.build_gcn_norm(sym=False)                      # column degrees: reverse-exchange of a ones column
    out["gcn_sym_false"] = sg3.gcn(x_own, be.f32(k)).cpu().numpy()
    out["gat_used_parts"] = bool(getattr(sg, "_gat_parts_cache", None))
    P.HUB_THRESHOLD, P.HUB_CHUNK = None, None
    results[rank] = out
    return out


def _entry(rank, world, port, use_gpu, skew, path, rounds=None, partitioned=False, hub_threshold=None, self_halo=False,
           env=None):
    os.environ.update(env or {})
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if use_gpu:
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    run_checks(rank, world, use_gpu, skew, res, rounds=rounds, partitioned=partitioned, hub_threshold=hub_threshold,
               self_halo=self_halo)
    np.save(os.path.join(path, "rank{}.npy".format(rank)), np.array([res[rank]], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    """A rendezvous port the kernel just handed out as free (bound to port 0 on the loopback interface, as bench.py's
    self_launch does) — never a number drawn blindly from the ephemeral range, where any other socket may already live."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn(world, use_gpu, skew, path, port, rounds=None, partitioned=False, hub_threshold=None, self_halo=False, env=None):
    import torch.multiprocessing as mp
    mp.spawn(_entry, args=(world, port, use_gpu, skew, path, rounds, partitioned, hub_threshold, self_halo, env),
             nprocs=world, join=True)
    return [np.load(os.path.join(path, "rank{}.npy".format(r)), allow_pickle=True)[0] for r in range(world)]


def reference(skew):
    from oracle import tfg_oracle as oracle
    ei, x, w, k, b = make_inputs(skew=skew)
    return {
        "gcn": oracle.gcn(x, ei, w, k, b, "relu"),
        "gcn_nokernel": oracle.gcn(x, ei, w, None),
        "mean": oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.mean_reducer, oracle.identity_updater),
        "max": oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.max_reducer, oracle.identity_updater),
        "sum_unweighted": oracle.aggregate_neighbors(x, ei, None, oracle.identity_mapper, oracle.sum_reducer,
                                                     oracle.identity_updater),
        "gcn_left_improved_unweighted": oracle.gcn(x, ei, None, k, norm="left", improved=True),
        "gcn_sym_false": oracle.gcn(x, ei, w, k, sym=False),
        **_sage_reference(oracle, x, ei, w),
        "gat": oracle.gat(x, ei, gat_weights()[0], gat_weights()[3], "relu", gat_weights()[1], gat_weights()[3], "relu",
                          gat_weights()[2], b[:8], "relu", num_heads=2),
    }


def _sage_reference(oracle, x, ei, w):
    rs = np.random.Generator(np.random.PCG64(99))
    f = x.shape[1]
    ks, kn = (rs.standard_normal((f, 5)) * 0.3).astype(np.float32), (rs.standard_normal((f, 5)) * 0.3).astype(np.float32)
    kmlp, bmlp = (rs.standard_normal((f, 8)) * 0.3).astype(np.float32), (rs.standard_normal(8) * 0.1).astype(np.float32)
    kn2, b10 = (rs.standard_normal((8, 5)) * 0.3).astype(np.float32), (rs.standard_normal(10) * 0.1).astype(np.float32)
    return {
        "sage_mean": oracle.mean_graph_sage(x, ei, w, ks, kn, b10, "relu"),
        "sage_sum_add": oracle.sum_graph_sage(x, ei, w, ks, kn, b10[:5], None, concat=False),
        "sage_max_pool": oracle.max_pool_graph_sage(x, ei, w, ks, kmlp, kn2, bmlp, b10, "relu"),
        "sage_mean_pool": oracle.mean_pool_graph_sage(x, ei, w, ks, kmlp, kn2, bmlp, b10, "relu"),
        "sage_mean_pool_add": oracle.mean_pool_graph_sage(x, ei, w, ks, kmlp, kn2, bmlp, b10[:5], "relu", concat=False),
    }


def check_against_reference(parts, skew, assert_parity):
    ref = reference(skew)
    parts = sorted(parts, key=lambda p: p["lo"])
    assert parts[0]["lo"] == 0 and all(a["hi"] == b["lo"] for a, b in zip(parts, parts[1:]))
    for key, full in ref.items():
        if key not in parts[0]:
            continue
        got = np.concatenate([p[key] for p in parts], axis=0)
        if key == "sage_max_pool":
            # an isolated node keeps float32 lowest() through the next GEMM (graph_sage.py:269): sums of +-1e38 terms
            # overflow or cancel in fp32 (inf - inf -> nan -> relu -> 0), so those rows are excluded from the comparison
            huge = np.abs(full).max(axis=1) > 1e30
            got, full = got[~huge], full[~huge]
        assert_parity(got, full, what="sharded " + key)
    return parts


# ----------------------------------------------------------------------------------------------------------------------
# training: sharded backward (reverse halo exchange + owner-side accumulate) and the weight-gradient all-reduce
# ----------------------------------------------------------------------------------------------------------------------
def _loss_coef(n, u):
    return (np.sin(np.arange(n * u, dtype=np.float64) * 0.37).reshape(n, u) + 1.5).astype(np.float32)


def run_training(rank, world, use_gpu, skew, rounds=None, num_splits=None, hub_threshold=None, self_halo=False,
                 transport=None):
    from tf_geometric_amd.dist.sharded import ShardedGraph
    import tf_geometric_amd.plan as P
    # hub_threshold: force the chunked long-row paths of the backward kernels (transposed local pass, GAT / max gradients
    # on the shard's rectangular plan) on this small graph
    P.HUB_THRESHOLD, P.HUB_CHUNK = hub_threshold, (None if hub_threshold is None else max(2, hub_threshold // 2))
    ei, x, w, k, b = make_inputs(skew=skew)
    n = x.shape[0]
    if use_gpu:
        backend = None
    else:
        from cpu_backend import NumpyBackend
        backend = NumpyBackend()
    group = dist.group.WORLD if dist.is_initialized() else None
    sg = ShardedGraph.from_global(ei, n, edge_weight=w, group=group, backend=backend, rounds=rounds, transport=transport,
                                  self_halo_rows=(n // world) // 3 if self_halo else None)
    be = sg.backend
    sg.build_gcn_norm()
    x_own = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
    kernel = be.f32(k).requires_grad_(True)
    bias = be.f32(b).requires_grad_(True)
    coef = be.f32(_loss_coef(n, k.shape[1])[sg.own_lo:sg.own_hi])
    out = sg.gcn_trainable(x_own, kernel, bias, torch.relu)
    (out * coef).sum().backward()
    local_dk = kernel.grad.detach().clone()
    sg.all_reduce_gradients([kernel, bias])
    res = {"lo": sg.own_lo, "hi": sg.own_hi, "out": out.detach().cpu().numpy(), "dx": x_own.grad.cpu().numpy(),
           "dk": kernel.grad.cpu().numpy(), "db": bias.grad.cpu().numpy(), "dk_local": local_dk.cpu().numpy()}
    # mean aggregation (GraphSAGE's reduce) through the same backward
    x2 = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
    m = sg.aggregate_trainable(x2, 1)
    (m * be.f32(_loss_coef(n, x.shape[1])[sg.own_lo:sg.own_hi])).sum().backward()
    res["dx_mean"] = x2.grad.cpu().numpy()
    # max aggregation and the fused attention: differentiable halo table + the single-GPU backward on the shard's plan
    x3 = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
    mx = sg.aggregate_trainable(x3, 2, w=None)
    (mx * be.f32(_loss_coef(n, x.shape[1])[sg.own_lo:sg.own_hi])).sum().backward()
    res["out_max"], res["dx_max"] = mx.detach().cpu().numpy(), x3.grad.cpu().numpy()
    # the same max at 36 columns (x tiled three times): wide enough for the HIP backend's tracked span-by-span forward
    # (own-source span under the exchange, one sub-span per round, merged in the kernel epilogue); every column block must
    # reproduce the 12-column result
    xw = be.f32(np.tile(x[sg.own_lo:sg.own_hi], (1, 3))).requires_grad_(True)
    mw = sg.aggregate_trainable(xw, 2, w=None)
    (mw * be.f32(np.tile(_loss_coef(n, x.shape[1])[sg.own_lo:sg.own_hi], (1, 3)))).sum().backward()
    res["out_max_wide"], res["dx_max_wide"] = mw.detach().cpu().numpy(), xw.grad.cpu().numpy()
    gw = [be.f32(a).requires_grad_(True) for a in gat_train_weights()]
    x4 = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
    og = sg.gat_trainable(x4, gw[0], gw[1], 1, gw[2], gw[3], 1, gw[4], gw[5], torch.relu, GAT_HEADS)
    (og * be.f32(_loss_coef(n, og.shape[1])[sg.own_lo:sg.own_hi])).sum().backward()
    sg.all_reduce_gradients(gw)
    res["out_gat"], res["dx_gat"] = og.detach().cpu().numpy(), x4.grad.cpu().numpy()
    res["dw_gat"] = [t.grad.cpu().numpy() for t in gw]
    # mean-pool GraphSAGE (trainable): the mean is linear, so both projections run on the owner and ku-wide rows travel
    mw_ = [be.f32(a).requires_grad_(True) for a in pool_weights()]
    x6 = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
    om = sg.pool_graph_sage_trainable(x6, mw_[0], mw_[1], mw_[2], mw_[3], mw_[4], act=1, concat=True, op=1)
    (om * be.f32(_loss_coef(n, om.shape[1])[sg.own_lo:sg.own_hi])).sum().backward()
    sg.all_reduce_gradients(mw_)
    res["out_meanpool"], res["dx_meanpool"] = om.detach().cpu().numpy(), x6.grad.cpu().numpy()
    res["dw_meanpool"] = [t.grad.cpu().numpy() for t in mw_]
    if not skew:     # (the skewed graph has rows without in-edges: their float-lowest maxima overflow the next GEMM in fp32)
        pw = [be.f32(a).requires_grad_(True) for a in pool_weights()]
        x5 = be.f32(x[sg.own_lo:sg.own_hi]).requires_grad_(True)
        op_ = sg.pool_graph_sage_trainable(x5, pw[0], pw[1], pw[2], pw[3], pw[4], act=1, concat=True, op=2)
        (op_ * be.f32(_loss_coef(n, op_.shape[1])[sg.own_lo:sg.own_hi])).sum().backward()
        sg.all_reduce_gradients(pw)
        res["out_pool"], res["dx_pool"] = op_.detach().cpu().numpy(), x5.grad.cpu().numpy()
        res["dw_pool"] = [t.grad.cpu().numpy() for t in pw]
    # static input features: the halo is exchanged once, later aggregations run without any exchange
    st = sg.prepare_static_features(be.f32(x[sg.own_lo:sg.own_hi]))
    calls = {"n": 0}
    real_start = sg.exchange_start

    def counting_start(table):
        calls["n"] += 1
        return real_start(table)
    sg.exchange_start = counting_start
    a1 = sg.aggregate_static(st, 0, w=sg.norm_w, self_coef=sg.self_coef)
    a2 = sg.aggregate_static(st, 1)
    sg.exchange_start = real_start
    t_ref = sg.alloc_table(x.shape[1])
    sg.own_rows(t_ref).copy_(be.f32(x[sg.own_lo:sg.own_hi]))
    res["static_sum"], res["static_sum_ref"] = a1.cpu().numpy(), sg.aggregate(t_ref, 0, w=sg.norm_w, self_coef=sg.self_coef).cpu().numpy()
    res["static_mean"], res["static_mean_ref"] = a2.cpu().numpy(), sg.neighbor_reduce(be.f32(x[sg.own_lo:sg.own_hi]), 1).cpu().numpy()
    res["static_exchanges"] = calls["n"]
    res["counters"] = dict(getattr(sg, "counters", {}))
    if num_splits:
        chunked = sg.aggregate_chunked(be.f32(x[sg.own_lo:sg.own_hi]), num_splits, w=sg.norm_w, self_coef=sg.self_coef,
                                       bias=be.f32(np.arange(x.shape[1], dtype=np.float32) * 0.01), act=1)
        table = sg.alloc_table(x.shape[1])
        sg.own_rows(table).copy_(be.f32(x[sg.own_lo:sg.own_hi]))
        whole = sg.aggregate(table, 0, w=sg.norm_w, self_coef=sg.self_coef,
                             bias=be.f32(np.arange(x.shape[1], dtype=np.float32) * 0.01), act=1)
        res["chunked"], res["whole"] = chunked.cpu().numpy(), whole.cpu().numpy()
        res["chunk_table_floats"], res["full_table_floats"] = sg.last_chunk_table_floats, int(table.numel())
    return res


def _train_entry(rank, world, port, use_gpu, skew, path, rounds, num_splits, hub_threshold=None, self_halo=False, env=None):
    os.environ.update(env or {})
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if use_gpu:
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run_training(rank, world, use_gpu, skew, rounds, num_splits, hub_threshold, self_halo=self_halo)
    np.save(os.path.join(path, "train{}.npy".format(rank)), np.array([res], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def spawn_training(world, use_gpu, skew, path, port, rounds=None, num_splits=None, hub_threshold=None, self_halo=False,
                   env=None):
    import torch.multiprocessing as mp
    mp.spawn(_train_entry, args=(world, port, use_gpu, skew, path, rounds, num_splits, hub_threshold, self_halo, env),
             nprocs=world, join=True)
    return [np.load(os.path.join(path, "train{}.npy".format(r)), allow_pickle=True)[0] for r in range(world)]


def training_reference(skew):
    """Single-process float64 autograd over the oracle's normalised adjacency: what tf.GradientTape yields."""
    from oracle import tfg_oracle as oracle
    ei, x, w, k, b = make_inputs(skew=skew)
    n = x.shape[0]
    nei, nw = oracle.gcn_norm_adj(ei, w, n)
    A = torch.zeros((n, n), dtype=torch.float64)
    A.index_put_((torch.from_numpy(nei[0].astype(np.int64)), torch.from_numpy(nei[1].astype(np.int64))),
                 torch.from_numpy(nw.astype(np.float64)), accumulate=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    kt = torch.tensor(k, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    out = torch.relu(A @ (xt @ kt) + bt)
    (out * torch.from_numpy(_loss_coef(n, k.shape[1]).astype(np.float64))).sum().backward()
    W = torch.zeros((n, n), dtype=torch.float64)
    W.index_put_((torch.from_numpy(ei[0].astype(np.int64)), torch.from_numpy(ei[1].astype(np.int64))),
                 torch.from_numpy(w.astype(np.float64)), accumulate=True)
    deg = torch.from_numpy(np.maximum(np.bincount(ei[0], minlength=n), 1).astype(np.float64))
    x2 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    ((W @ x2) / deg[:, None] * torch.from_numpy(_loss_coef(n, x.shape[1]).astype(np.float64))).sum().backward()
    ref = {"out": out.detach().numpy(), "dx": xt.grad.numpy(), "dk": kt.grad.numpy(), "db": bt.grad.numpy(),
           "dx_mean": x2.grad.numpy()}
    # max aggregation, GAT layer, max-pool GraphSAGE layer: float64 torch autograd over the edge list as the reference
    # composes them (map_reduce.py:31-42; gat.py:40-122; graph_sage.py:228-287) — scatter amax shares tied gradients
    rows, cols = torch.from_numpy(ei[0].astype(np.int64)), torch.from_numpy(ei[1].astype(np.int64))
    lowest = float(np.finfo(np.float32).min)

    def seg_max(msg):
        base = torch.full((n, msg.shape[1]), lowest, dtype=torch.float64)
        return base.scatter_reduce(0, rows.unsqueeze(1).expand_as(msg), msg, reduce="amax", include_self=True)
    x3 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    mx = seg_max(x3[cols])
    (mx * torch.from_numpy(_loss_coef(n, x.shape[1]).astype(np.float64))).sum().backward()
    ref["out_max"], ref["dx_max"] = mx.detach().numpy(), x3.grad.numpy()
    gw = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in gat_train_weights()]
    x4 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    H = GAT_HEADS
    r2, c2 = torch.cat([rows, torch.arange(n)]), torch.cat([cols, torch.arange(n)])
    Q, K, V = torch.relu(x4 @ gw[0] + gw[1]), torch.relu(x4 @ gw[2] + gw[3]), x4 @ gw[4]
    d, dv = Q.shape[1] // H, V.shape[1] // H
    sc = (Q.view(n, H, d)[r2] * K.view(n, H, d)[c2]).sum(-1) / float(np.sqrt(d))
    mxs = torch.full((n, H), -1e300, dtype=torch.float64).scatter_reduce(0, r2.unsqueeze(1).expand_as(sc), sc.detach(),
                                                                         reduce="amax")
    ex = torch.exp(sc - mxs[r2])
    den = torch.zeros((n, H), dtype=torch.float64).index_add(0, r2, ex)
    alpha = ex / (den[r2] + 1e-8)
    og = torch.zeros((n, H, dv), dtype=torch.float64).index_add(0, r2, alpha.unsqueeze(-1) * V.view(n, H, dv)[c2])
    og = torch.relu(og.reshape(n, H * dv) + gw[5])
    (og * torch.from_numpy(_loss_coef(n, og.shape[1]).astype(np.float64))).sum().backward()
    ref["out_gat"], ref["dx_gat"], ref["dw_gat"] = og.detach().numpy(), x4.grad.numpy(), [t.grad.numpy() for t in gw]
    mw_ = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in pool_weights()]
    x6 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    hm6 = torch.relu(x6 @ mw_[1] + mw_[3])
    cnt = torch.zeros(n, dtype=torch.float64).index_add_(0, rows, torch.ones(rows.shape[0], dtype=torch.float64)).clamp(min=1)
    mean6 = torch.zeros((n, hm6.shape[1]), dtype=torch.float64).index_add(0, rows, hm6[cols]) / cnt.unsqueeze(1)
    om = torch.relu(torch.cat([x6 @ mw_[0], mean6 @ mw_[2]], 1) + mw_[4])
    (om * torch.from_numpy(_loss_coef(n, om.shape[1]).astype(np.float64))).sum().backward()
    ref["out_meanpool"], ref["dx_meanpool"] = om.detach().numpy(), x6.grad.numpy()
    ref["dw_meanpool"] = [t.grad.numpy() for t in mw_]
    if not skew:
        pw = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in pool_weights()]
        x5 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        hm = torch.relu(x5 @ pw[1] + pw[3])
        op_ = torch.relu(torch.cat([x5 @ pw[0], seg_max(hm[cols]) @ pw[2]], 1) + pw[4])
        (op_ * torch.from_numpy(_loss_coef(n, op_.shape[1]).astype(np.float64))).sum().backward()
        ref["out_pool"], ref["dx_pool"], ref["dw_pool"] = op_.detach().numpy(), x5.grad.numpy(), [t.grad.numpy() for t in pw]
    return ref


def check_training_extras(parts, ref, assert_parity):
    """max aggregation / GAT layer / max-pool SAGE layer of the sharded training path against the float64 reference."""
    parts = sorted(parts, key=lambda p: p["lo"])
    if "out_max_wide" in parts[0]:
        assert_parity(np.concatenate([p["out_max_wide"] for p in parts]), np.tile(ref["out_max"], (1, 3)), tol=2e-5,
                      what="sharded trainable max forward, 36 columns")
        assert_parity(np.concatenate([p["dx_max_wide"] for p in parts]), np.tile(ref["dx_max"], (1, 3)), tol=1e-4,
                      what="sharded max d/dx, 36 columns")
    for key in ("max", "gat", "pool", "meanpool"):
        if "out_" + key not in ref:
            continue
        assert_parity(np.concatenate([p["out_" + key] for p in parts]), ref["out_" + key], tol=2e-5,
                      what="sharded trainable {} forward".format(key))
        assert_parity(np.concatenate([p["dx_" + key] for p in parts]), ref["dx_" + key], tol=1e-4,
                      what="sharded {} d/dx".format(key))
        if "dw_" + key in ref:
            for p in parts:
                for i, (got, want) in enumerate(zip(p["dw_" + key], ref["dw_" + key])):
                    assert_parity(got, want, tol=2e-4, what="all-reduced {} weight gradient {}".format(key, i))


def _agree_entry(rank, world, port, path, fail_rank):
    """tests/test_dist_gloo.py::test_strict_transport_failure_is_agreed: the strict C-ABI transport on a gloo control
    group.  fail_rank None: nothing is patched (on a box without a GPU every rank fails its local preconditions);
    fail_rank r: only rank r's library load fails.  Every rank must raise TfgxDistUnavailable — nobody waits in RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tf_geometric_amd.dist import transport as T
    from tf_geometric_amd import _lib as L
    if fail_rank is not None:
        class _FakeLib(object):                       # the preconditions "pass" everywhere except on fail_rank
            pass

        def fake_load():
            if rank == fail_rank:
                raise L.TfgxError("injected: libtfgx_dist.so missing on this rank")
            return _FakeLib()
        T.load_dist_library = fake_load
        L.require_gpu = lambda: None
        L.device = lambda: torch.device("cpu")
        torch.cuda.Stream = lambda device=None: None
    msg = None
    try:
        T.get_transport(None, None, "tfgx_dist")
    except T.TfgxDistUnavailable as ex:
        msg = str(ex)
    with open(os.path.join(path, "agree{}.txt".format(rank)), "w") as fh:
        fh.write(msg or "NO ERROR")
    dist.barrier()
    dist.destroy_process_group()


def spawn_agree(world, path, port, fail_rank):
    import torch.multiprocessing as mp
    mp.spawn(_agree_entry, args=(world, port, path, fail_rank), nprocs=world, join=True)
    return [open(os.path.join(path, "agree{}.txt".format(r))).read() for r in range(world)]
