# coding=utf-8
"""bench.py — aggregated edges/s + HBM roofline of the GCN propagation (segment-sum) hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload products|arxiv|cora|tiny]

A "step" is one pass of the hot path over the whole graph: out = A_hat @ h with the GCN-normalised adjacency
(weighted gather - scale - segment-sum into destination nodes, the implicit self-loop term included; h is the
[N, F] float32 feature matrix, resident in HBM before the timed region) — i.e. tfg.layers.GCN(use_kernel=False,
use_bias=False) on the cached plan.  Default workload: the ogbn-products-shaped graph BASELINE.json's target is
quoted on (N = 2.4 M, E = 123 M, F = 100).

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME graph is sharded by destination-node range,
halo source rows are exchanged as an all-to-all-v over RCCL and overlapped with the local-edge pass
("scaling": "strong").

Prints ONE JSON line (rank 0). `roofline` prices the dominant kernel (seg_reduce_kernel) with ALGORITHMIC bytes
(SURVEY.md §8d): B_alg = E_agg*(4F + 8) + N*4F + 4(N+1), E_agg = E + N, against the 8 TB/s HBM3E peak.
`cpu_baseline` times the C restatement of the reference path (oracle/tfg_oracle.c, "port") on the host cores over
a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # MI355X HBM3E peak, B/s (MI355X_MICROARCH.md)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="products")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--extras", action="store_true", help="also time GEMM+aggregation layers (GCN/SAGE/GAT)")
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args()


def b_alg(e_agg, n, f, weighted=True):
    return e_agg * (4 * f + 4 + (4 if weighted else 0)) + n * 4 * f + 4 * (n + 1)


def usable_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota.  (On the GPU boxes of
    this pool the mask shows all 256 hardware threads of the 2 x EPYC 9575F host but cpu.max grants 16 CPUs; 256
    OpenMP threads under that quota run 6x slower than 16 — measured 37 vs 226 M edges/s.)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                    # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        try:                                                          # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                cores = max(1, min(cores, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(x_np, ei_np, w_np, self_coef_np, n, f, budget_edges):
    """C port of gather -> gcn_mapper -> unsorted_segment_sum (oracle/tfg_oracle.c), all host cores, on the
    destination rows [0, n_s) of the same graph (full source range, so the gather locality is the job's)."""
    lib_path = os.path.join(ROOT, "oracle", "libtfg_oracle.so")
    lib = ctypes.CDLL(lib_path)
    fn = lib.tfgo_aggregate_csr_f32
    fn.restype = ctypes.c_int
    cores = usable_cores()
    e_total = ei_np.shape[1]
    frac = min(1.0, float(budget_edges) / max(e_total, 1))
    n_s = max(1, int(n * frac))
    keep = ei_np[0] < n_s
    diag = np.arange(n_s, dtype=np.int32)            # the appended self-loop edges (add_diag, gcn.py:77)
    row = np.ascontiguousarray(np.concatenate([ei_np[0][keep], diag]))
    col = np.ascontiguousarray(np.concatenate([ei_np[1][keep], diag]))
    w = np.ascontiguousarray(np.concatenate([w_np[keep], self_coef_np[:n_s]]))
    out = np.empty((n_s, f), dtype=np.float32)
    P = ctypes.c_void_p
    # NUMA: spread the feature matrix over the host's memory controllers by first-touch in parallel (untimed)
    x_cpu = np.empty_like(x_np)
    lib.tfgo_parallel_copy_f32(P(x_cpu.ctypes.data), P(x_np.ctypes.data), ctypes.c_int64(x_np.size), ctypes.c_int(cores))
    x_np = x_cpu
    # plan (untimed, like the GPU leg's): stable sort by destination -> row_ptr / col / w in CSR order
    order = np.argsort(row, kind="stable")
    col = np.ascontiguousarray(col[order])
    w = np.ascontiguousarray(w[order])
    row_ptr = np.zeros(n_s + 1, dtype=np.int32)
    np.cumsum(np.bincount(row, minlength=n_s), out=row_ptr[1:])

    def run():
        rc = fn(P(x_np.ctypes.data), ctypes.c_int64(f), P(row_ptr.ctypes.data), P(col.ctypes.data),
                P(w.ctypes.data), ctypes.c_int64(n_s), ctypes.c_int64(f), ctypes.c_int(0), P(out.ctypes.data),
                ctypes.c_int64(f), ctypes.c_int(cores))
        assert rc == 0

    run()  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": row.shape[0] / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "dst rows [0,{}) of the same graph: {} edges x F={} , full source range, {:.2f} s per pass, "
                      "tfgo_aggregate_csr_f32 (row-sorted edges, OpenMP, {} threads; sort untimed)".format(
                          n_s, int(row.shape[0]), f, dt, cores)}, out, n_s


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node {} bench.py --gpus {}".format(
            args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    import tf_geometric_amd as tfg
    from tf_geometric_amd import synthetic
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj

    L.require_gpu()
    n, e_req, f = synthetic.WORKLOADS[args.workload]
    ei_np = synthetic.synthetic_edges(n, e_req, seed=args.seed)
    e = int(ei_np.shape[1])
    x_np = synthetic.synthetic_features(n, f, seed=args.seed + 1)
    w_np = np.ones(e, dtype=np.float32)      # Graph default edge_weight (data/graph.py:53-56)

    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") always, except for the single-GPU plumbing check of this code path
        # (TFGX_BENCH_BACKEND=gloo with all ranks on one device; rows are then staged through the host)
        dist.init_process_group(os.environ.get("TFGX_BENCH_BACKEND", "nccl"))
        from tf_geometric_amd.dist.sharded import ShardedGraph
        t0 = time.perf_counter()
        sg = ShardedGraph.from_global(ei_np, n, edge_weight=None, group=dist.group.WORLD)
        sg.build_gcn_norm()
        table = sg.alloc_table(f)                       # [own rows | halo rows]; own rows resident before timing
        sg.own_rows(table).copy_(L.as_f32(x_np[sg.own_lo:sg.own_hi]))
        out = torch.empty((sg.n_own, f), dtype=torch.float32, device=table.device)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t0

        def step():   # pack -> RCCL all-to-all-v (async) || local-source pass -> halo-source pass
            return sg.gcn_propagate(table, out=out)

        def barrier():
            dist.barrier()
    else:
        t0 = time.perf_counter()
        ei = L.as_i32(ei_np)
        adj = tfg.SparseMatrix(ei, None, [n, n])
        cache = {}
        normed = gcn_norm_adj(adj, cache=cache)
        x = L.as_f32(x_np)
        out = torch.empty((n, f), dtype=torch.float32, device=x.device)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t0
        from tf_geometric_amd.plan import segment_reduce

        def step():
            return segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out)

        def barrier():
            pass

    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        res = step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1) / args.steps

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        t = torch.tensor([ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ev_ms = float(t.item())

    ms_per_step = wall * 1e3 / args.steps
    e_agg = e + n
    line = {
        "metric": "aggregated edges/sec + achieved HBM GB/s, GCN layer",
        "value": e * args.steps / wall,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",          # the SAME graph at every N: total work fixed, per-GPU work shrinks
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "{}-shaped GCN propagation A_hat@h (weighted segment-sum + implicit self-loops): "
                               "N={} E={} F={}".format(args.workload, n, e, f),
                   "nodes": n, "edges": e, "features": f, "edges_aggregated": e_agg,
                   "partition": "single GPU" if world == 1 else "dst-range x{} + RCCL halo all-to-all-v".format(world)},
        "plan_build_s": plan_s,
    }

    if rank == 0 and world > 1:
        # whole job: algorithmic bytes of the full graph over the step time (exchange included) vs N x 8 TB/s
        bytes_alg = b_alg(e_agg, n, f, weighted=True)
        achieved = bytes_alg / (ms_per_step * 1e-3)
        line["roofline"] = {"bound": "hbm", "kernel": "seg_reduce_kernel (local pass + halo pass) + RCCL all-to-all-v",
                            "achieved": achieved / 1e9, "peak": world * HBM_PEAK / 1e9, "unit": "GB/s",
                            "frac": achieved / (world * HBM_PEAK), "traffic": None,
                            "algorithmic_bytes_per_launch": bytes_alg, "step_ms": ms_per_step,
                            "halo_rows_received_rank0": sg.n_halo, "halo_bytes_received_rank0": sg.n_halo * f * 4,
                            "edges_rank0": sg.num_edges, "rows_rank0": sg.n_own}
    if rank == 0 and world == 1:
        bytes_alg = b_alg(e_agg, n, f, weighted=True)
        achieved = bytes_alg / (ev_ms * 1e-3)
        # compulsory lower bound: every array touched exactly once
        bytes_min = 8 * e_agg + 4 * (n + 1) + 2 * n * 4 * f
        kname = "seg_reduce_kernel<4,32,1,sum,weighted>" if f == 100 else "seg_reduce_kernel (sum, weighted; F={})".format(f)
        line["roofline"] = {"bound": "hbm", "kernel": kname,
                            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK, "traffic": None,
                            "algorithmic_bytes_per_launch": bytes_alg, "kernel_ms": ev_ms,
                            "bytes_per_edge": 4 * f + 8,
                            "compulsory_bytes_per_launch": bytes_min,
                            "frac_compulsory": bytes_min / (ev_ms * 1e-3) / HBM_PEAK}
        # 128-byte line requests: what actually bounds a gather.  A 4F-byte row at a 4F-byte stride
        # touches ceil((offset mod 128 + 4F) / 128) lines; the ceiling is the measured random-span fetch rate of
        # tools/line_rate_probe.cpp over a table of this size, when that profile is committed.
        row_bytes = 4 * f
        offs = sorted({(row_bytes * i) % 128 for i in range(128)})
        lines_per_row = sum(-(-(o + row_bytes) // 128) for o in offs) / float(len(offs))
        line["roofline"]["row_lines_per_launch"] = e_agg * lines_per_row
        line["roofline"]["row_lines_per_s"] = e_agg * lines_per_row / (ev_ms * 1e-3)
        probe_path = os.path.join(ROOT, "profiles", "r01_line_rate_probe_916MiB.jsonl")
        if args.workload == "products" and os.path.exists(probe_path):
            best = 0.0
            with open(probe_path) as fh:
                for ln in fh:
                    rec = json.loads(ln)
                    if rec.get("probe") == "random_spans":
                        best = max(best, rec["G_lines_per_s"] * 1e9)
            line["roofline"]["random_line_ceiling_per_s"] = best
            line["roofline"]["frac_of_random_line_ceiling"] = line["roofline"]["row_lines_per_s"] / best
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE;
        # see profiles/): a property of kernel + workload, cannot be sampled from inside this process.
        pmc_path = os.path.join(ROOT, "profiles", "r01_{}_pmc.json".format(args.workload))
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            line["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
            line["roofline"]["traffic_source"] = "profiles/" + os.path.basename(pmc_path)
            line["roofline"]["traffic_GBps"] = pmc["traffic_bytes_per_launch"] / (ev_ms * 1e-3) / 1e9
            line["roofline"]["frac_traffic"] = pmc["traffic_bytes_per_launch"] / (ev_ms * 1e-3) / HBM_PEAK
        # The same pass with the source features in the static-feature layout (SplitRows + edge-resident tail columns,
        # DESIGN.md §2.1): what layer 0 runs from the second epoch on.  Reported NEXT TO the headline, never as it: the
        # layout is derived from the feature values once (build_ms), like the plan is derived from the edges.
        from tf_geometric_amd.plan import SplitRows
        if SplitRows.wanted(n, f):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rows = SplitRows.from_dense(x).with_edge_tail(normed.plan)
            torch.cuda.synchronize()
            build_ms = (time.perf_counter() - t1) * 1e3
            out2 = torch.empty_like(out)
            for _ in range(args.warmup):
                segment_reduce(normed.plan, rows, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                segment_reduce(normed.plan, rows, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out2)
            e1.record()
            torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / args.steps
            line["static_feature_layout"] = {
                "what": "same launch, x stored as main[N,96] + tail[N,4] + the tail columns of each edge's source row "
                        "streamed next to col/w (3 line requests per gathered row instead of 4)",
                "kernel_ms": ms2, "edges_per_s": e / (ms2 * 1e-3), "frac_of_hbm_peak_algorithmic": bytes_alg / (ms2 * 1e-3) / HBM_PEAK,
                "build_ms_once_per_feature_matrix": build_ms, "extra_bytes": int(rows.edge_tail.numel() * 4),
                "bit_identical_to_headline_output": bool(torch.equal(out2, res))}
            et_path = os.path.join(ROOT, "profiles", "r01_{}_edge_tail_pmc.json".format(args.workload))
            if os.path.exists(et_path):          # measured HBM-side bytes of this launch (rocprofv3 PMC passes)
                with open(et_path) as fh:
                    line["static_feature_layout"]["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
                line["static_feature_layout"]["traffic_source"] = "profiles/" + os.path.basename(et_path)
            del rows, out2
        if not args.no_cpu_baseline:
            budget = {"products": 61_500_000}.get(args.workload, e)
            base, cpu_out, n_s = cpu_baseline(x_np, ei_np, normed_w_host(normed, ei_np, n),
                                              normed.self_coef.cpu().numpy(), n, f, budget)
            line["cpu_baseline"] = base
            gpu_rows = res[:n_s].cpu().numpy()
            line["parity_vs_cpu_port_max_abs_err"] = float(np.abs(gpu_rows - cpu_out).max())
            line["speedup_vs_cpu_baseline"] = line["value"] / base["value"]
    if args.extras and world == 1 and rank == 0:
        line["extras"] = extras(tfg, L, synthetic, x, ei, n, e, f, cache)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def normed_w_host(normed, ei_np, n):
    """GCN-normalised weights in the caller's edge order for the CPU leg (so both legs do the same arithmetic)."""
    w_csr = normed.w_csr.cpu().numpy()
    perm = normed.plan.perm.cpu().numpy()
    w = np.empty_like(w_csr)
    w[perm] = w_csr
    return w


def _time(fn, steps=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def extras(tfg, L, synthetic, x, ei, n, e, f, cache):
    """Whole-layer timings on the same graph (ms): GEMM + aggregation through the layer API."""
    res = {"note": "layers called repeatedly with the SAME feature tensor and cache switch to the static-feature layout "
                   "from the second call on (DESIGN.md §2.1); *_first_call_ms keys time the plain dense layout"}
    w1 = torch.ones(e, dtype=torch.float32, device=x.device)
    gcn = tfg.layers.GCN(256, activation=tfg.relu)
    plain = {k: v for k, v in cache.items()}           # same plan / normalised adjacency, but forget the feature tensor

    def first_call(layer, inputs):
        c = dict(plain)
        c.pop("tfgx_static_rows", None)
        return layer(inputs, cache=c)

    gcn([x, ei], cache=cache)
    plain = {k: v for k, v in cache.items() if k != "tfgx_static_rows"}
    res["gcn_layer_F{}_to_256_first_call_ms".format(f)] = _time(lambda: first_call(gcn, [x, ei]))
    res["gcn_layer_F{}_to_256_ms".format(f)] = _time(lambda: gcn([x, ei], cache=cache))
    sage = tfg.layers.MeanGraphSage(256)
    sage([x, ei, w1], cache=cache)
    plain = {k: v for k, v in cache.items() if k != "tfgx_static_rows"}
    res["mean_sage_layer_units256_first_call_ms"] = _time(lambda: first_call(sage, [x, ei, w1]))
    res["mean_sage_layer_units256_ms"] = _time(lambda: sage([x, ei, w1], cache=cache))
    mp = tfg.layers.MaxPoolGraphSage(64)
    res["maxpool_sage_layer_units64_ms"] = _time(lambda: mp([x, ei, w1], cache=cache))
    gat = tfg.layers.GAT(64, attention_units=8, num_heads=8, activation=tfg.relu)
    res["gat_layer_H8_A8_U64_ms"] = _time(lambda: gat([x, ei], cache=cache))
    # 2-layer GCN (F -> 256 -> 40, BASELINE configs[1] model): eager launches vs one hipGraph replay
    g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)

    def two_layer(xx):
        return g1([g0([xx, ei], cache=cache), ei], cache=cache)

    res["gcn_2layer_eager_ms"] = _time(lambda: two_layer(x))
    cap = tfg.CapturedForward(two_layer, x)
    res["gcn_2layer_hipgraph_ms"] = _time(lambda: cap.graph.replay())
    # training step of one GCN layer (forward + backward through the autograd kernels)
    gt = tfg.layers.GCN(256, activation=tfg.relu)
    gt._maybe_build([x])
    gt.trainable(True)

    def train_step():
        for p_ in gt.parameters():
            p_.grad = None
        gt([x, ei], cache=cache).sum().backward()

    res["gcn_layer_fwd_bwd_ms"] = _time(train_step, steps=5, warmup=2)
    # one full-batch training step of the 2-layer model (F -> 256 -> 40): forward, cross-entropy on 10 % of the nodes,
    # backward through both layers (aggregation on the transposed plan, GEMM gradients), Adam update
    t0l, t1l = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)
    t0l.trainable(True)
    t1l.trainable(True)
    with torch.no_grad():
        t1l([t0l([x, ei], cache=cache), ei], cache=cache)
    opt = torch.optim.Adam(t0l.parameters() + t1l.parameters(), lr=1e-2)
    idx = torch.arange(0, n, 10, device=x.device)
    labels = torch.randint(0, 40, (int(idx.shape[0]),), device=x.device)

    def full_step():
        opt.zero_grad(set_to_none=True)
        logits = t1l([t0l([x, ei], cache=cache), ei], cache=cache)
        torch.nn.functional.cross_entropy(logits[idx], labels).backward()
        opt.step()

    res["gcn_2layer_train_step_ms"] = _time(full_step, steps=5, warmup=2)
    from tf_geometric_amd.plan import gemm_bias_act
    k = L.as_f32(synthetic.glorot_uniform(f, 256))
    ms = _time(lambda: gemm_bias_act(x, k))
    res["gemm_{}x{}x256_ms".format(n, f)] = ms
    res["gemm_tflops"] = 2.0 * n * f * 256 / (ms * 1e-3) / 1e12
    return res


if __name__ == "__main__":
    main()
