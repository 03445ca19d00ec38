# coding=utf-8
"""bench.py — aggregated edges/s + HBM roofline of the GCN propagation (segment-sum) hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload products|arxiv|cora|tiny]

A "step" is one pass of the hot path over the whole graph: out = A_hat @ h with the GCN-normalised adjacency
(weighted gather - scale - segment-sum into destination nodes, the implicit self-loop term included; h is the
[N, F] float32 feature matrix, resident in HBM before the timed region) — i.e. tfg.layers.GCN(use_kernel=False,
use_bias=False) on the cached plan.  Default workload: the ogbn-products-shaped graph BASELINE.json's target is
quoted on (N = 2.4 M, E = 123 M, F = 100).

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME graph is sharded by destination-node range
(every rank generates only ITS stripe of the edge list: ShardedGraph.from_partitioned), halo source rows are
exchanged as an all-to-all-v over RCCL and overlapped with the local-edge pass ("scaling": "strong").  The line then
carries per-rank halo bytes and the exchange / local-pass / halo-pass times measured alone, so a scaling curve can be
read (roofline.per_rank, roofline.overlap_frac).

Prints ONE JSON line (rank 0).
  roofline      prices the dominant kernel with ALGORITHMIC bytes: B_alg = E_agg*(4F + 8) + N*4F +
                4(N+1), E_agg = E + N, against the 8 TB/s HBM3E peak; `kernel` is the symbol the dispatcher reports for
                this launch (tfgx_segment_reduce_describe); kernel_ms comes from HIP events on the launch stream.
                `traffic` is IMPORTED from the committed rocprofv3 PMC profile of the same command (a counter pass
                cannot run inside this process) and is priced with that profile's own kernel time.
  cpu_baseline  the C restatement of the reference path (oracle/tfg_oracle.c, "port": CSR + OpenMP — a STRONG CPU
                baseline) on a bounded sample; `cpu_baseline_op_for_op` next to it times the reference's formulation AS
                WRITTEN (nn/kernel/map_reduce.py:62-70: gather -> multiply -> unsorted_segment_sum, three [E, F]
                tensors) in torch-CPU on the same cores.
  rmat          the same launch on an R-MAT graph of the same size (skewed in-degrees, hub path) — beside the uniform
                graph, never instead of it.
  static_feature_layout   the same launch with the features in the opt-in static layout (tfg.prepare_static_features).
  configs       (N = 1, round 5) the OTHER BASELINE.json configs, bounded to seconds: C2 arxiv-shaped 2-layer GCN forward +
                training step, C3 Reddit-shaped GAT(64, H8, A8) forward / forward + backward / attention alone, C4
                GraphSAGE(256, concat) mean and max-pool layers with the dominant reduce broken out, the width sweep
                F = 128 ... 512, two GEMMs beside torch.matmul — each with ms, the dispatched kernel and an algorithmic fraction.
  watchdog      phase times of the run; a phase that makes no progress for TFGX_BENCH_WATCHDOG_S seconds (default 120) ends the
                process with ONE JSON line carrying "error" and exit code 3 (every rank runs one).
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

# Multi-process GPU work on the MI355X boxes needs dmabuf IPC (the host driver has no legacy IPC mode): without it RCCL's
# bring-up fails inside hipIpcGetMemHandle.  The ROCr runtime reads the variable at its initialisation, so it is defaulted
# before torch is imported; the ranks self_launch() starts inherit it.  (tf_geometric_amd.dist does the same at import.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402

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
    p.add_argument("--no-rmat", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the block that times BASELINE configs[1..3] beside the headline")
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


def _job_token():
    """Names ONE job: the launcher's pid, its start time (field 22 of /proc/<pid>/stat: a recycled pid — the launcher is often
    pid 1 or a small recycled pid inside a container — gets another start time) and the rendezvous port."""
    ppid = os.getppid()
    start = "0"
    try:
        with open("/proc/{}/stat".format(ppid)) as fh:
            start = fh.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return "{}_{}_{}".format(ppid, start, os.environ.get("MASTER_PORT", "0"))


def _error_lock_path():
    import tempfile
    return os.path.join(tempfile.gettempdir(), "tfgx_bench_error_{}.lock".format(_job_token()))


class Watchdog(object):
    """One timer thread per rank.  Every phase of the run is announced with stage(name, limit_s); when a phase makes no
    progress within its limit the thread reports and ends the process (exit code 3): ONE JSON line carrying
    "error" on the saved stdout descriptor (N > 1: the first rank whose timer fires), every rank a sentence on stderr.  A blocked ncclCommInitRank, grouped exchange
    or gloo barrier cannot be interrupted from Python any other way, and without this the driver would see nothing but its
    own 1800 s timeout.  Limits: TFGX_BENCH_WATCHDOG_S (default 120 s per phase; the phases that generate / sort / time
    on the host get a multiple of it)."""

    def __init__(self, line_fd, rank, world):
        self.line_fd, self.rank, self.world = line_fd, rank, world
        self.base = float(os.environ.get("TFGX_BENCH_WATCHDOG_S", "120"))
        self.name, self.limit, self.t0 = "start", self.base, time.monotonic()
        self.history = []
        self._stop = False
        self._thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self._thread.start()

    def stage(self, name, factor=1.0):
        now = time.monotonic()
        self.history.append((self.name, round(now - self.t0, 3)))
        self.name, self.limit, self.t0 = name, self.base * factor, now

    def stop(self):
        self.stage("done")
        self._stop = True
        if self.world > 1 and self.rank == 0:        # a normal exit leaves nothing behind (nor does it find a stale file: _job_token)
            try:
                os.unlink(_error_lock_path())
            except OSError:
                pass

    def _run(self):
        while not self._stop:
            time.sleep(0.5)
            waited = time.monotonic() - self.t0
            if not self._stop and waited > self.limit:
                msg = "bench.py watchdog: rank {} of {} made no progress in phase '{}' for {:.0f} s (limit {:.0f} s)".format(
                    self.rank, self.world, self.name, waited, self.limit)
                try:
                    sys.stderr.write(msg + "; phases so far: {}\n".format(self.history))
                    sys.stderr.flush()
                    # exactly ONE line on stdout: the first rank whose timer fires writes it (the launcher ends the other
                    # ranks as soon as one exits, so waiting for rank 0 could lose the reason).  The ranks of a job are
                    # children of one launcher process: an O_EXCL file named after it elects the writer.
                    first = True
                    if self.world > 1:
                        try:
                            os.close(os.open(_error_lock_path(), os.O_CREAT | os.O_EXCL | os.O_WRONLY))
                        except FileExistsError:
                            first = False
                        except OSError:
                            first = self.rank == 0            # no writable temp dir: rank 0 reports
                    if first:
                        line = {"metric": "aggregated edges/sec + achieved HBM GB/s, GCN layer", "value": None,
                                "unit": "edges/s", "n_gpus": self.world, "error": msg, "phase": self.name, "reported_by_rank": self.rank,
                                "phases_completed": self.history,
                                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
                        os.write(self.line_fd, (json.dumps(line) + "\n").encode())
                finally:
                    os._exit(3)


# ----------------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1 only; bounded samples of the same workload; the oracle library is used here and only here)
# ----------------------------------------------------------------------------------------------------------------------
def _sample_rows(ei_np, w_np, self_coef_np, n, budget_edges):
    """Destination rows [0, n_s) of the same graph with ALL their in-edges (full source range, so the gather locality
    is the job's), the appended self-loop edges included (add_diag, gcn.py:77)."""
    e_total = ei_np.shape[1]
    frac = min(1.0, float(budget_edges) / max(e_total, 1))
    n_s = max(1, int(n * frac))
    keep = ei_np[0] < n_s
    diag = np.arange(n_s, dtype=np.int32)
    row = np.ascontiguousarray(np.concatenate([ei_np[0][keep], diag]))
    col = np.ascontiguousarray(np.concatenate([ei_np[1][keep], diag]))
    w = np.ascontiguousarray(np.concatenate([w_np[keep], self_coef_np[:n_s]]))
    return row, col, w, n_s


def cpu_baseline_port(x_np, ei_np, w_np, self_coef_np, n, f, budget_edges):
    """C port of gather -> gcn_mapper -> unsorted_segment_sum as a CSR loop (oracle/tfg_oracle.c), all usable cores."""
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libtfg_oracle.so"))
    fn = lib.tfgo_aggregate_csr_f32
    fn.restype = ctypes.c_int
    cores = usable_cores()
    row, col, w, n_s = _sample_rows(ei_np, w_np, self_coef_np, n, budget_edges)
    out = np.empty((n_s, f), dtype=np.float32)
    P = ctypes.c_void_p
    # NUMA: spread the feature matrix over the host's memory controllers by first-touch in parallel (untimed)
    x_cpu = np.empty_like(x_np)
    lib.tfgo_parallel_copy_f32(P(x_cpu.ctypes.data), P(x_np.ctypes.data), ctypes.c_int64(x_np.size), ctypes.c_int(cores))
    # plan (untimed, like the GPU leg's): stable sort by destination -> row_ptr / col / w in CSR order
    order = np.argsort(row, kind="stable")
    col_s = np.ascontiguousarray(col[order])
    w_s = np.ascontiguousarray(w[order])
    row_ptr = np.zeros(n_s + 1, dtype=np.int32)
    np.cumsum(np.bincount(row, minlength=n_s), out=row_ptr[1:])

    def run():
        rc = fn(P(x_cpu.ctypes.data), ctypes.c_int64(f), P(row_ptr.ctypes.data), P(col_s.ctypes.data),
                P(w_s.ctypes.data), ctypes.c_int64(n_s), ctypes.c_int64(f), ctypes.c_int(0), P(out.ctypes.data),
                ctypes.c_int64(f), ctypes.c_int(cores))
        assert rc == 0

    run()  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        if time.perf_counter() - t0 > 8.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": row.shape[0] / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "formulation": "CSR loop over destination rows (row-sorted edges, OpenMP): a STRONG CPU baseline, not how "
                           "the reference computes",
            "sample": "dst rows [0,{}) of the same graph: {} edges x F={} , full source range, {:.2f} s per pass, "
                      "tfgo_aggregate_csr_f32 ({} threads; sort untimed)".format(n_s, int(row.shape[0]), f, dt, cores)
            }, out, n_s


def cpu_baseline_op_for_op(x_np, ei_np, w_np, self_coef_np, n, f, budget_edges):
    """The reference's formulation AS WRITTEN (nn/kernel/map_reduce.py:62-70 with gcn_mapper, nn/conv/gcn.py:221-222):
    repeated_x = gather(x, row); neighbor_x = gather(x, col); msg = neighbor_x * w[:, None];
    out = unsorted_segment_sum(msg, row, n) — three [E, F] tensors materialised — in torch-CPU on the same cores
    (TensorFlow itself is not installable in this image)."""
    cores = usable_cores()
    torch.set_num_threads(cores)
    row, col, w, n_s = _sample_rows(ei_np, w_np, self_coef_np, n, budget_edges)
    xt = torch.from_numpy(x_np)
    rt, ct, wt = torch.from_numpy(row.astype(np.int64)), torch.from_numpy(col.astype(np.int64)), torch.from_numpy(w)

    def run():
        repeated_x = xt.index_select(0, rt.clamp(max=n - 1))       # map_reduce.py:62 (dead for gcn_mapper, still executed)
        neighbor_x = xt.index_select(0, ct)                         # :63
        msg = neighbor_x * wt.unsqueeze(1)                          # gcn.py:222
        out = torch.zeros((n_s, f), dtype=torch.float32).index_add_(0, rt, msg)     # :70 unsorted_segment_sum
        del repeated_x
        return out

    out = run()  # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        out = run()
        reps += 1
        if time.perf_counter() - t0 > 8.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": row.shape[0] / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "formulation": "op for op as the reference executes: gather x2 -> multiply -> unsorted_segment_sum "
                           "(torch-CPU index_select / mul / index_add_), three [E,F] float32 tensors materialised",
            "sample": "dst rows [0,{}) of the same graph: {} edges x F={}, {:.2f} s per pass, {} threads".format(
                n_s, int(row.shape[0]), f, dt, cores)}, out.numpy()


# ----------------------------------------------------------------------------------------------------------------------
# R-MAT variant: (a,b,c,d) = (0.57,0.19,0.19,0.05), generated on the GPU
# ----------------------------------------------------------------------------------------------------------------------
def kernel_source_sha():
    """sha256[:16] of the headline kernel's sources — the same recipe tools/make_pmc_json.py stamps into profiles/*_pmc.json."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("tf_geometric_amd/csrc/tfgx_reduce.hip", "tf_geometric_amd/csrc/tfgx_common.h"):
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _event_time(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def rmat_line(tfg, L, n, e, f, x, steps, warmup, seed):
    from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj
    from tf_geometric_amd.plan import segment_reduce
    from tf_geometric_amd.synthetic import rmat_edges
    ei = rmat_edges(n, e, seed, x.device)
    e_r = int(ei.shape[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    normed = gcn_norm_adj(tfg.SparseMatrix(ei, None, [n, n]), sym=True)
    plan = normed.plan
    hub = plan.hub_info()
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    out = torch.empty((n, f), dtype=torch.float32, device=x.device)
    ms = _event_time(lambda: segment_reduce(plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out),
                     steps, warmup)
    bytes_alg = b_alg(e_r + n, n, f)
    deg = plan.in_degree()
    return {"what": "same launch on an R-MAT graph (a,b,c,d)=(0.57,0.19,0.19,0.05) of the same N / E / F "
                    "(generated on the GPU, both directions emitted); long rows take the chunked hub path",
            "edges": e_r, "kernel_ms": ms, "edges_per_s": e_r / (ms * 1e-3),
            "roofline": hbm_roof(bytes_alg, ms),
            "max_in_degree": int(deg.max().item()), "empty_rows": int((deg == 0).sum().item()),
            "hub_rows": 0 if hub is None else int(hub[0].shape[0]), "hub_threshold": int(getattr(plan, "hub_threshold", 0)),
            "plan_build_s": plan_s}


# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    does) and become that launcher — rank 0 of the children prints the one JSON line on this process's stdout."""
    import socket
    with socket.socket() as sock:                 # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    # ONE JSON line on stdout, nothing else: the C++ layers underneath (gloo's "[Gloo] Rank 0 is connected to ..." banner,
    # RCCL's version banner) print to fd 1, so fd 1 is pointed at stderr for the whole run and the line is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if args.gpus > 1 and world == 1:
        os.dup2(line_fd, 1)                       # the children inherit the real stdout
        self_launch(args)                         # does not return
    wd = Watchdog(line_fd, rank, world)
    hang_at = os.environ.get("TFGX_BENCH_TEST_HANG", "")      # tests only: block inside the named phase, as a stuck collective would

    def maybe_hang(phase):
        if hang_at and hang_at == phase and (os.environ.get("TFGX_BENCH_TEST_HANG_RANK", str(rank)) == str(rank)):
            wd.stage("TEST HANG in '{}'".format(phase))
            time.sleep(10 ** 6)

    maybe_hang("start")
    wd.stage("HIP initialisation")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    n_dev = torch.cuda.device_count()
    # N > 1: one rank per GPU, halo rows over the repo's own RCCL communicator (transport "tfgx_dist", asked for BY NAME:
    # if any rank cannot bring it up every rank exits non-zero — a scaling line is never carried by anything else).
    # The one exception is the single-GPU plumbing check of this code path, which must be requested explicitly:
    # TFGX_BENCH_BACKEND=gloo puts all ranks on the visible device(s) and stages rows through the host (transport "torch").
    plumbing = os.environ.get("TFGX_BENCH_BACKEND", "") == "gloo"
    torch.cuda.set_device(local_rank % n_dev)

    import tf_geometric_amd as tfg
    from tf_geometric_amd import synthetic
    from tf_geometric_amd import _lib as L
    from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj

    L.require_gpu()
    n, e_req, f = synthetic.WORKLOADS[args.workload]
    diag = None
    gen_s = None

    if world > 1:
        import torch.distributed as dist
        # CONTROL PLANE = a gloo group (rendezvous, the 128-byte ncclUniqueId, barriers, the max-over-ranks of the timings):
        # host tensors only.  DATA PLANE = tfgx_dist's ncclComm_t: the only RCCL communicator in the process.
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        wd.stage("gloo rendezvous (control plane)")
        dist.init_process_group("gloo")
        wd.stage("device census over the control plane")
        # one physical device per rank?  (ranks may each see all GPUs, or one each through *_VISIBLE_DEVICES: compare PCI ids)
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        mine = "{}:{:04x}:{:02x}:{:02x}".format(os.uname().nodename, getattr(props, "pci_domain_id", 0),
                                               getattr(props, "pci_bus_id", local_rank % n_dev), getattr(props, "pci_device_id", 0))
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        if len(set(devs)) < world and not plumbing:
            raise SystemExit("bench.py --gpus {}: {} distinct GPU(s) behind the {} ranks; RCCL needs one device per rank (for a "
                             "plumbing check of the N > 1 code path on fewer GPUs set TFGX_BENCH_BACKEND=gloo)".format(
                                 world, len(set(devs)), world))
        from tf_geometric_amd.dist.sharded import ShardedGraph, HipBackend
        from tf_geometric_amd.dist.transport import get_transport, ipc_mode_note
        # the data plane first, as a phase of its own: a hang inside ncclCommInitRank or the first grouped send / receive is
        # then reported BY NAME by the watchdog within its limit
        wd.stage("RCCL bring-up over tfgx_dist (ncclGetUniqueId, ncclCommInitRank, self-check rows)" if not plumbing
                 else "host-staged transport (plumbing mode)")
        maybe_hang("bring-up")
        transport = get_transport(dist.group.WORLD, HipBackend(), "torch" if plumbing else "tfgx_dist")
        wd.stage("stripe generation (host)", 3)
        t0 = time.perf_counter()
        # every rank GENERATES only its stripe of the edge list (block-seeded generator: the union over the ranks is the same
        # edge multiset at every N) and from_partitioned routes each edge to its destination's owner — nothing edge-sized
        # is replicated on a host or on a GPU
        stripe = synthetic.synthetic_edge_stripe(n, e_req, seed=args.seed, stripe=rank, num_stripes=world)
        gen_s = time.perf_counter() - t0
        cnt = torch.tensor([int(stripe.shape[1])], dtype=torch.int64)
        dist.all_reduce(cnt)
        e = int(cnt.item())
        wd.stage("shard plan: degree all-reduce, edge routing all-to-all-v, CSR + halo plan", 3)
        sg = ShardedGraph.from_partitioned(stripe, n, group=dist.group.WORLD, transport=transport)
        del stripe
        wd.stage("sharded GCN normalisation (one 1-column halo exchange)")
        sg.build_gcn_norm()
        wd.stage("own feature rows (host generation + copy)", 3)
        table = sg.alloc_table(f)                       # [own rows | halo rows]; own rows resident before timing
        x_own = synthetic.synthetic_feature_rows(n, f, seed=args.seed + 1, row_lo=sg.own_lo, row_hi=sg.own_hi)
        sg.own_rows(table).copy_(L.as_f32(x_own))
        del x_own
        out = torch.empty((sg.n_own, f), dtype=torch.float32, device=table.device)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t0

        def step():   # pack -> RCCL all-to-all-v (async) || local-source pass -> halo-source pass
            return sg.gcn_propagate(table, out=out)

        def barrier():
            dist.barrier()

        def max_over_ranks(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
    else:
        wd.stage("input generation (host)", 3)
        t0 = time.perf_counter()
        ei_np = synthetic.synthetic_edge_stripe(n, e_req, seed=args.seed)     # all blocks: the same graph the N > 1 runs shard
        e = int(ei_np.shape[1])
        x_np = synthetic.synthetic_feature_rows(n, f, seed=args.seed + 1)
        gen_s = time.perf_counter() - t0
        wd.stage("host -> device copies, CSR plan, GCN normalisation")
        t0 = time.perf_counter()
        ei = L.as_i32(ei_np)
        x = L.as_f32(x_np)
        torch.cuda.synchronize()
        h2d_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        adj = tfg.SparseMatrix(ei, None, [n, n])
        cache = {}
        normed = gcn_norm_adj(adj, cache=cache)
        cache["tfgx_csr_plan"] = adj.plan           # one CSR plan per graph, shared by every layer given this cache
        out = torch.empty((n, f), dtype=torch.float32, device=x.device)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t0
        # the same again with the code objects loaded and the allocator warm: what a second graph costs, split with HIP
        # events into the CSR build (sort by destination) and the normalisation
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t1 = time.perf_counter()
        ev[0].record()
        adj2 = tfg.SparseMatrix(ei, None, [n, n])
        _ = adj2.plan
        ev[1].record()
        normed2 = gcn_norm_adj(adj2, cache={})
        ev[2].record()
        torch.cuda.synchronize()
        plan_rebuild_s = time.perf_counter() - t1
        plan_rebuild_split = {"csr_build_ms_events": ev[0].elapsed_time(ev[1]), "gcn_norm_ms_events": ev[1].elapsed_time(ev[2])}
        del adj2, normed2
        from tf_geometric_amd.plan import segment_reduce

        def step():
            return segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out)

        def barrier():
            pass

    cold_ms = None
    wd.stage("first step (N > 1: first halo exchange on the RCCL communicator + the per-round passes)")
    maybe_hang("first-step")
    for i in range(args.warmup):
        if i == 1:
            wd.stage("warm-up steps")
        if i == 0:          # the very first launch after plan build: cold caches / TLBs, code object load
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
        res = step()
        if i == 0:
            c1.record()
            torch.cuda.synchronize()
            cold_ms = c0.elapsed_time(c1)
    torch.cuda.synchronize()
    wd.stage("barrier in front of the timed loop")
    barrier()
    torch.cuda.synchronize()
    wd.stage("timed loop ({} steps)".format(args.steps), 2)
    import gc
    gc.collect()
    gc.disable()            # as timeit does: a full cyclic collection of a torch process is a 40 ms host pause
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
    gc.enable()
    ev_ms = ev0.elapsed_time(ev1) / args.steps

    if world > 1:
        import torch.distributed as dist
        wd.stage("max over ranks + per-rank diagnostics (exchange / passes alone)", 2)
        wall = max_over_ranks(wall)
        ev_ms = max_over_ranks(ev_ms)
        # everything below is commentary beside the headline: a failure in it (the same code on every rank, so the same
        # failure on every rank) is reported inside the line instead of costing the run its number
        diag, static_shard = None, None
        try:
            diag = shard_diagnostics(sg, table, out, f, L, dist, max(3, min(args.steps, 10)))
        except Exception as ex:                                         # noqa: BLE001
            static_shard = {"error": "shard_diagnostics: {!r}".format(ex)}
        # beside the headline (which exchanges the halo every step, as any hidden layer must): layer 0 with its input
        # features declared static — halo exchanged once, shard table in the edge-resident-tail layout, no exchange per step
        if diag is not None:
            wd.stage("static shard layout (halo exchanged once)", 2)
            try:
                st = sg.prepare_static_features(sg.own_rows(table))
                dist.barrier()
                ms_static = max_over_ranks(_event_time(
                    lambda: sg.aggregate_static(st, L.SUM, w=sg.norm_w, self_coef=sg.self_coef, out=out),
                    max(3, min(args.steps, 10)), 2))
                static_shard = {"what": "layer 0 after ShardedGraph.prepare_static_features: halo exchanged once, no exchange "
                                        "per step, shard table in the static layout where the width calls for it",
                                "step_ms_max_over_ranks": ms_static, "edges_per_s": e / (ms_static * 1e-3),
                                "bytes_rank0": int(st["bytes"]),
                                "layout_rank0": "edge_tail" if st["split"] is not None else "dense"}
                del st
            except Exception as ex:                                     # noqa: BLE001
                static_shard = {"error": "static feature layout: {!r}".format(ex)}

    wd.stage("commentary beside the headline (static layout, R-MAT, CPU legs, configs)", 5)
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
                   "partition": "single GPU" if world == 1 else
                                "dst-range x{} + {} halo all-to-all-v".format(world, "HOST-STAGED (plumbing check)" if plumbing else "RCCL")},
        "plan_build_s": plan_s,
        "plan_build_s_is": ("first CSR plan + GCN normalisation of this process on device-resident edges: includes code-object "
                            "loads and first allocations (host perf_counter around a synchronize); inputs generated in "
                            "input_generation_s and copied in h2d_copy_s, both outside it") if world == 1 else
                           ("this rank's stripe generation + from_partitioned (degree all-reduce, edge routing all-to-all-v, "
                            "CSR + halo plan) + GCN normalisation + own feature rows to the device (host perf_counter)"),
        "input_generation_s": gen_s,
    }
    if world == 1:
        line["h2d_copy_s"] = h2d_s
        line["plan_rebuild_on_device_s"] = plan_rebuild_s
        line["plan_rebuild_split"] = plan_rebuild_split
    if world > 1:
        # which exchange implementation carried the halo rows: "tfgx_dist" = the C ABI of include/tfgx_dist.h (in-process
        # ncclComm_t, grouped ncclSend / ncclRecv on a second HIP stream) — the product path on RCCL, and the only thing a
        # scaling line may be carried by; "torch" only in the explicitly requested one-GPU plumbing check.
        # rccl_ranks = ncclCommCount of that communicator (0: no RCCL communicator, plumbing mode).
        line["config"]["transport"] = sg.transport.name
        line["config"]["control_plane"] = "gloo (host tensors: rendezvous, barriers, max-over-ranks)"
        line["config"]["rccl_ranks"] = sg.transport.comm_info()[0] if hasattr(sg.transport, "comm_info") else 0
        line["config"]["devices_visible"] = n_dev
        line["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        if ipc_mode_note():
            line["config"]["ipc_mode_note"] = ipc_mode_note()
        if plumbing:
            line["config"]["plumbing_check"] = ("TFGX_BENCH_BACKEND=gloo: {} ranks on {} device(s), rows staged through the "
                                                "host — exercises the N > 1 code path, NOT a scaling number".format(world, n_dev))
        elif line["config"]["rccl_ranks"] != world or sg.transport.name != "tfgx_dist":
            raise SystemExit("bench.py: the halo exchange is not on a {}-rank tfgx_dist communicator ({}, {} ranks)".format(
                world, sg.transport.name, line["config"]["rccl_ranks"]))

    if rank == 0 and world > 1:
        # whole job: algorithmic bytes of the full graph over the step time (exchange included) vs N x 8 TB/s
        bytes_alg = b_alg(e_agg, n, f, weighted=True)
        achieved = bytes_alg / (ms_per_step * 1e-3)
        serial, hideable = 0.0, 0.0
        if diag:
            slow = max(diag, key=lambda d: d["exchange_ms"] + d["local_pass_ms"] + d["halo_pass_ms"])
            serial = slow["exchange_ms"] + slow["local_pass_ms"] + slow["halo_pass_ms"]
            hideable = min(slow["exchange_ms"], slow["local_pass_ms"] + slow["halo_pass_ms"])
        line["roofline"] = {"bound": "hbm", "kernel": "seg_reduce_kernel (local-source pass + {} halo-round passes) "
                                                      "+ RCCL all-to-all-v in {} rounds".format(sg.rounds, sg.rounds),
                            "achieved": achieved / 1e9, "peak": world * HBM_PEAK / 1e9, "unit": "GB/s",
                            "frac": achieved / (world * HBM_PEAK), "traffic": None,
                            "algorithmic_bytes_per_launch": bytes_alg, "step_ms": ms_per_step,
                            "per_rank": diag, "static_feature_layout": static_shard,
                            "overlap_frac": max(0.0, min(1.0, (serial - ms_per_step) / hideable)) if hideable > 0 else None,
                            "overlap_note": "exchange_ms / local_pass_ms / halo_pass_ms are each measured ALONE "
                                            "(barrier-separated) after the timed loop; overlap_frac = (their sum on the "
                                            "slowest rank - step_ms) / min(exchange, passes): 1 = the exchange is fully "
                                            "hidden, 0 = fully serial"}
    if rank == 0 and world == 1:
        bytes_alg = b_alg(e_agg, n, f, weighted=True)
        achieved = bytes_alg / (ev_ms * 1e-3)
        # compulsory lower bound: every array touched exactly once
        bytes_min = 8 * e_agg + 4 * (n + 1) + 2 * n * 4 * f
        kname = segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out,
                               describe=True)
        # per-launch distribution (SURVEY.md §8d timing protocol): 20 individually timed launches after the timed loop
        singles = []
        for _ in range(20):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            step()
            a1.record()
            torch.cuda.synchronize()
            singles.append(a0.elapsed_time(a1))
        singles.sort()
        line["roofline"] = {"bound": "hbm", "kernel": kname,
                            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK, "traffic": None,
                            "algorithmic_bytes_per_launch": bytes_alg, "kernel_ms": ev_ms,
                            "kernel_ms_median_of_20": singles[len(singles) // 2], "kernel_ms_min": singles[0],
                            "kernel_ms_max": singles[-1], "cold_first_launch_ms": cold_ms,
                            "cache_state": "L2 / MALL are NOT flushed between launches (the 0.96 GB feature matrix is "
                                           "3.7x the 256 MB MALL; measured L2 hit rate 2 %)",
                            "aggregated_edges_per_s": e_agg / (ev_ms * 1e-3),
                            "bytes_per_edge": 4 * f + 8,
                            "compulsory_bytes_per_launch": bytes_min,
                            "frac_compulsory": bytes_min / (ev_ms * 1e-3) / HBM_PEAK}
        # 128-byte line requests: what actually bounds a gather.  A 4F-byte row at a 4F-byte stride
        # touches ceil((offset mod 128 + 4F) / 128) lines.
        row_bytes = 4 * f
        offs = sorted({(row_bytes * i) % 128 for i in range(128)})
        lines_per_row = sum(-(-(o + row_bytes) // 128) for o in offs) / float(len(offs))
        line["roofline"]["row_lines_per_launch"] = e_agg * lines_per_row
        line["roofline"]["row_lines_per_s"] = e_agg * lines_per_row / (ev_ms * 1e-3)
        # HBM-side bytes per launch: IMPORTED from the committed rocprofv3 PMC profile of this command (FETCH_SIZE x2 on
        # gfx950 + WRITE_SIZE, separate --pmc passes; see profiles/).  Priced with the PROFILE's own kernel time — it
        # was taken on another box of the pool — never with this run's.
        sha_now = kernel_source_sha()
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc_path = os.path.join(ROOT, "profiles", "{}_{}_pmc.json".format(tag, args.workload))
            if os.path.exists(pmc_path):
                with open(pmc_path) as fh:
                    pmc = json.load(fh)
                if pmc.get("kernel_source_sha16") != sha_now:
                    # the profile describes ANOTHER build of the kernel: keep the line, drop the number
                    line["roofline"]["traffic"] = None
                    line["roofline"]["traffic_is"] = ("not reported: profiles/{} was measured on kernel sources {} but the "
                                                      "tree holds {} (re-run tools/profile_round.sh)".format(
                                                          os.path.basename(pmc_path), pmc.get("kernel_source_sha16"), sha_now))
                    break
                line["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                line["roofline"]["traffic_is"] = "imported from profiles/{} (not measured in this run)".format(
                    os.path.basename(pmc_path))
                pms = pmc.get("kernel_ms")
                if pms:
                    line["roofline"]["traffic_profile_kernel_ms"] = pms
                    # FETCH_SIZE / WRITE_SIZE count the L2s' fabric-side requests: Infinity-Cache hits are INCLUDED, so this is
                    # the rate of requests leaving the L2s, not DRAM bandwidth (the part streams ~6.3 TB/s from HBM)
                    line["roofline"]["l2_fabric_side_GBps_in_profile"] = pmc["traffic_bytes_per_launch"] / (pms * 1e-3) / 1e9
                    line["roofline"]["traffic_counts"] = ("L2 fabric-side read + write requests (FETCH_SIZE x 2 on gfx950 + "
                                                          "WRITE_SIZE): Infinity-Cache hits included")
                line["roofline"]["traffic_over_algorithmic"] = pmc["traffic_bytes_per_launch"] / float(bytes_alg)
                break
        # The same pass with the features declared static (tfg.prepare_static_features: SplitRows + edge-resident tail
        # columns) — what layer 0 of a model runs in every epoch once the caller has opted in.  Reported
        # NEXT TO the headline, never as it: the layout is derived from the feature values (build_ms, extra bytes).
        # build cost, two readings: the FIRST call on the host clock (includes a fresh 2.9 GB hipMalloc through torch's
        # allocator — 5 to 60 ms depending on the box) and a second build, allocator warm, between HIP events (the kernels)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        info = tfg.prepare_static_features(x, normed.plan, cache)
        torch.cuda.synchronize()
        build_first_ms = (time.perf_counter() - t1) * 1e3
        tfg.release_static_features(cache)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        info = tfg.prepare_static_features(x, normed.plan, cache)
        b1.record()
        torch.cuda.synchronize()
        build_ms = b0.elapsed_time(b1)
        if info["layout"] == "edge_tail":
            from tf_geometric_amd.plan import static_rows
            rows = static_rows(x, normed.plan, cache)
            out2 = torch.empty_like(out)
            fn2 = lambda: segment_reduce(normed.plan, rows, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out2)   # noqa: E731
            ms2 = _event_time(fn2, args.steps, args.warmup)
            line["static_feature_layout"] = {
                "what": "same launch after tfg.prepare_static_features(x, ...): x stored as main[N,{}] + tail[N,{}] + "
                        "the tail columns of each edge's source row streamed next to col/w (3 line requests per "
                        "gathered row instead of 4)".format(info["f_main"], info["f_tail"]),
                "kernel": segment_reduce(normed.plan, rows, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef,
                                         out=out2, describe=True),
                "kernel_ms": ms2, "edges_per_s": e / (ms2 * 1e-3),
                "roofline": hbm_roof(bytes_alg, ms2),
                "build_ms_once_per_feature_matrix": build_ms,
                "build_ms_is": "HIP events around a second build (allocator warm): the layout kernels themselves",
                "build_first_call_ms_host_clock": build_first_ms, "layout_bytes": int(info["bytes"]),
                "bit_identical_to_headline_output": bool(torch.equal(out2, res))}
            for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):   # HBM-side bytes of this launch, imported like roofline.traffic
                et_path = os.path.join(ROOT, "profiles", "{}_{}_edge_tail_pmc.json".format(tag, args.workload))
                if os.path.exists(et_path):
                    with open(et_path) as fh:
                        et = json.load(fh)
                    if et.get("kernel_source_sha16") != kernel_source_sha():
                        break
                    line["static_feature_layout"]["traffic_imported"] = et["traffic_bytes_per_launch"]
                    line["static_feature_layout"]["traffic_source"] = "profiles/" + os.path.basename(et_path)
                    if et.get("kernel_ms"):
                        line["static_feature_layout"]["traffic_profile_kernel_ms"] = et["kernel_ms"]
                    break
            del rows, out2
        tfg.release_static_features(cache)
        # what a DROP-IN user gets without calling anything (plan.AUTO_STATIC_LAYOUT): the same propagation through the layer
        # API, tfg.layers.GCN(use_kernel=False)([x, edge_index], cache=cache), four calls in a row on a fresh sighting record —
        # call 1 reads x as it is, call 2 builds the layout (second sighting) and uses it, calls 3-4 run on it
        if info["layout"] == "edge_tail":
            from tf_geometric_amd import plan as P
            cache.pop("tfgx_static_seen", None)
            prop = tfg.layers.GCN(1, use_kernel=False, use_bias=False)
            calls = []
            for _ in range(4):
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                o_auto = prop([x, ei], cache=cache)
                c1.record()
                torch.cuda.synchronize()
                calls.append(c0.elapsed_time(c1))
            line["static_feature_layout"]["automatic_promotion"] = {
                "what": "GCN(use_kernel=False)([x, edge_index], cache=cache) x4, no opt-in call: ms per call (call 2 includes "
                        "the layout build)", "ms_per_call": calls,
                "promoted": cache.get("tfgx_static_rows", (None, None))[1] is not None,
                "bit_identical_to_headline_output": bool(torch.equal(o_auto, res)),
                "budget_bytes": int(P.static_layout_budget_bytes())}
            tfg.release_static_features(cache)
        if not args.no_rmat and n >= 100000:
            line["rmat"] = rmat_line(tfg, L, n, e_req, f, x, max(3, args.steps // 2), 2, args.seed + 7)
        if not args.no_cpu_baseline:
            budget = {"products": 61_500_000}.get(args.workload, e)
            w_host, sc_host = normed_w_host(normed, ei_np, n), normed.self_coef.cpu().numpy()
            base, cpu_out, n_s = cpu_baseline_port(x_np, ei_np, w_host, sc_host, n, f, budget)
            line["cpu_baseline"] = base
            gpu_rows = res[:n_s].cpu().numpy()
            line["parity_vs_cpu_port_max_abs_err"] = float(np.abs(gpu_rows - cpu_out).max())
            line["speedup_vs_cpu_baseline"] = line["value"] / base["value"]
            budget2 = {"products": 6_000_000}.get(args.workload, min(e, 6_000_000))     # 3 x [E_s, F] f32 host tensors
            base2, cpu_out2 = cpu_baseline_op_for_op(x_np, ei_np, w_host, sc_host, n, f, budget2)
            line["cpu_baseline_op_for_op"] = base2
            line["parity_vs_cpu_op_for_op_max_abs_err"] = float(np.abs(res[:cpu_out2.shape[0]].cpu().numpy() - cpu_out2).max())
            line["speedup_vs_cpu_op_for_op"] = line["value"] / base2["value"]
    if world == 1 and rank == 0 and not args.no_configs and n >= 100000:
        wd.stage("configs block (C2 arxiv GCN, C3 Reddit GAT, C4 GraphSAGE, GEMM)", 5)
        line["configs"] = baseline_configs(tfg, L, synthetic, x, ei, n, e, f, cache, args.seed)
    if args.extras and world == 1 and rank == 0:
        wd.stage("extras", 10)
        line["extras"] = extras(tfg, L, synthetic, x, ei, n, e, f, cache)
    line["watchdog"] = {"limit_s_per_phase": wd.base, "phases_s": wd.history + [(wd.name, round(time.monotonic() - wd.t0, 3))]}
    wd.stop()
    if rank == 0:
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def shard_diagnostics(sg, table, out, f, L, dist, reps):
    """Per rank, each measured ALONE with a barrier in front (so a slow peer is not counted as local time): the halo
    exchange (pack + all-to-all-v rounds + wait), the own-source pass, the halo-source passes on resident rows."""
    def timed(fn):
        ms = []
        for _ in range(reps):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        ms.sort()
        return ms[len(ms) // 2]

    kw = dict(w=sg.norm_w, out=out, exchange=False)
    exch = timed(lambda: sg.exchange_finish(sg.exchange_start(table)))
    local = timed(lambda: sg.aggregate(table, L.SUM, classes=[0], **kw))
    halo = timed(lambda: sg.aggregate(table, L.SUM, classes=list(range(1, sg.n_class)), self_coef=sg.self_coef, **kw))
    rpk = sg.rpk
    own_edges = int((rpk[1::sg.n_class] - rpk[0:-1:sg.n_class]).sum().item()) if sg.n_class > 1 else sg.num_edges
    mine = {"rank": sg.rank, "rows": sg.n_own, "edges": sg.num_edges, "own_source_edges": own_edges,
            "halo_rows_received": sg.n_halo, "halo_bytes_received": sg.n_halo * f * 4,
            "rows_sent": int(sum(sg.send_counts)), "bytes_sent": int(sum(sg.send_counts)) * f * 4,
            "rows_packed": int(sg.send_idx_packed.shape[0]), "peers_sent_whole_block_unpacked": int(sum(sg.dense_send)),
            "exchange_ms": exch, "local_pass_ms": local, "halo_pass_ms": halo,
            "exchange_GBps_received": sg.n_halo * f * 4 / (exch * 1e-3) / 1e9 if exch > 0 else None}
    gathered = [None] * sg.world
    dist.all_gather_object(gathered, mine)
    return gathered


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


def _time_ab(fn_a, fn_b, steps=20, warmup=3, rounds=5, detail=None):
    """Median per-step ms of two alternatives measured ALTERNATELY (a, b, a, b, ...): clock / power state drifts over
    a process's lifetime, so "first all of a, then all of b" charges the drift to whichever ran second (round 2's
    hipGraph-slower-than-eager reading was exactly that: a rocprofv3 trace of both shows the replay's kernels back to
    back with 8 us of total idle per step, profiles/r03_hipgraph_vs_eager.md)."""
    ta, tb = [], []
    for r in range(rounds):
        ta.append(_time(fn_a, steps, warmup if r == 0 else 1))
        tb.append(_time(fn_b, steps, warmup if r == 0 else 1))
    if detail is not None:           # every round as measured, in order: (a, b) pairs
        detail["rounds_ms"] = [[round(a, 4), round(b, 4)] for a, b in zip(ta, tb)]
    ta.sort()
    tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]


def _latency(fn, steps=20, warmup=3):
    """Mean wall time of fn() + synchronize, one step at a time (what a serving request or a training loop that reads the
    loss every step sees)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


def _fwd_bwd(layer, inputs, cache):
    def run():
        for p_ in layer.parameters():
            p_.grad = None
        layer(inputs, cache=cache).sum().backward()
    return run


def _entry(res, key, fn):
    """One entry of the configs block; a failure is reported inside the line, never at the cost of the headline."""
    try:
        res[key] = fn()
    except Exception as ex:                                            # noqa: BLE001
        res[key] = {"error": "{}: {}".format(type(ex).__name__, ex)}
    torch.cuda.synchronize()


L2_PEAK = 34.5e12          # bytes/s the eight 4 MiB L2s deliver together (MI355X_MICROARCH.md "L2 (per XCD)")
_PROBE_CACHE = {}


def line_rate_probe(table_mib):
    """Random aligned spans served by a table of `table_mib` MiB ON THIS BOX, NOW: tf_geometric_amd/lib/line_rate_probe
    (tools/line_rate_probe.cpp, a measurement utility built beside the library) -> {span_bytes: G lines/s}, or {"error": ...}.
    The roof of a gather whose table lives in a cache is the rate at which that cache serves random 128-byte lines, not a
    fraction of the HBM peak (VERDICT r5 missing #5)."""
    table_mib = max(1, int(round(table_mib)))
    if table_mib in _PROBE_CACHE:
        return _PROBE_CACHE[table_mib]
    exe = os.path.join(ROOT, "tf_geometric_amd", "lib", "line_rate_probe")
    res = {}
    try:
        import subprocess
        out = subprocess.run([exe, str(table_mib), "quick"], capture_output=True, text=True, timeout=60)
        for ln in out.stdout.splitlines():
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if d.get("probe") == "random_spans" and d.get("aligned"):
                res[int(d["span_bytes"])] = d["G_lines_per_s"]
        if not res:
            res = {"error": "no probe output (rc {}): {}".format(out.returncode, out.stderr[-200:])}
    except Exception as ex:                                            # noqa: BLE001
        res = {"error": "{}: {}".format(type(ex).__name__, ex)}
    _PROBE_CACHE[table_mib] = res
    return res


def roof(bound, achieved, peak, unit, **more):
    """One roofline object: achieved / peak in `unit`, frac = achieved / peak; `bound` names the resource the peak belongs to
    ("hbm": 8 TB/s; "l2": 34.5 TB/s of lines served by the XCDs' L2s; "mall": this box's probed random-line rate out of the
    Infinity Cache; "mfma": 155 TFLOP/s fp32)."""
    d = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": (achieved / peak) if peak else None}
    d.update(more)
    return d


def hbm_roof(bytes_alg, ms, **more):
    return roof("hbm", bytes_alg / (ms * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s", algorithmic_bytes=bytes_alg, **more)


def baseline_configs(tfg, L, synthetic, x, ei, n, e, f, cache, seed):
    """The OTHER BASELINE.json configs inside the driver-timed line — bounded (tens of seconds), after the
    timed loop, every entry with its time (HIP events, steps queued back to back), the kernel the dispatcher picks for the
    dominant launch and that launch's algorithmic fraction of the HBM peak (SURVEY.md §8d's B_alg).  Shapes:
    C2 demo/demo_gcn.py:22-23 on the arxiv-shaped graph (hidden 256, 40 classes); C3 demo/demo_gat.py:22 on the Reddit-shaped
    graph; C4 demo/demo_graph_sage.py:29-30 (units 256, concat) on the products-shaped graph this run already holds.
    The automatic static-layout promotion is OFF in here: every key times the tensors as they are."""
    from tf_geometric_amd import plan as P
    from tf_geometric_amd.plan import CsrPlan, segment_reduce, gemm_bias_act
    from tf_geometric_amd.nn.conv.gat import gat_attention
    res = {"what": "BASELINE.json configs[1..4] beside the headline: ms = HIP events over back-to-back launches; every entry "
                   "carries a `roofline` object for its dominant launch against the resource that bounds it — HBM (8 TB/s, "
                   "algorithmic bytes of SURVEY.md 8d) where the gathered table is far larger than the caches, the L2s (34.5 "
                   "TB/s of 128-byte lines) or the Infinity Cache (this box's probed random-line rate) where it is resident by "
                   "construction; promotion of static layouts switched off"}
    auto = P.AUTO_STATIC_LAYOUT
    P.AUTO_STATIC_LAYOUT = False
    dev = x.device
    t_start = time.perf_counter()

    # ---- C4: GraphSAGE mean / max-pool aggregators at the contract shape (units 256, concat) on the products-shaped graph
    w1 = torch.ones(e, dtype=torch.float32, device=dev)
    plan = CsrPlan.from_cache(ei, n, n, cache)

    def c4_mean():
        lay = tfg.layers.MeanGraphSage(256, activation=tfg.relu)
        before = P.FUSED_STATS["launches"]
        ms = _time(lambda: lay([x, ei, w1], cache=cache), steps=8, warmup=2)
        fused = P.FUSED_STATS["launches"] > before
        ms_agg = _time(lambda: segment_reduce(plan, x, L.MEAN, w_csr=w1), steps=8, warmup=2)
        tl = tfg.layers.MeanGraphSage(256, activation=tfg.relu)
        tl._maybe_build([x])
        tl.trainable(True)
        ms_t = _time(_fwd_bwd(tl, [x, ei, w1], cache), steps=4, warmup=2)
        balg = b_alg(e, n, f)
        return {"layer": "MeanGraphSage(256, concat=True) on [N={}, F={}], E={}".format(n, f, e),
                "forward_ms": ms, "fwd_bwd_ms": ms_t,
                "dominant_launch": ("tfgx_aggregate_gemm_f32 -> agg_gemm_kernel<32, true> (mean of w*x[col] at the INPUT width "
                                    "F={}, projected to 128 columns in the same launch)".format(f)) if fused else
                                   "tfgx_segment_reduce_f32 + tfgx_gemm_bias_act_f32",
                "aggregation_alone_kernel": segment_reduce(plan, x, L.MEAN, w_csr=w1, describe=True),
                "aggregation_alone_ms": ms_agg,
                "roofline": hbm_roof(balg, ms_agg, kernel="the aggregation alone (0.96 GB table, 3.7 x the Infinity Cache)"),
                "algorithmic_GBps_whole_layer": balg / (ms * 1e-3) / 1e9}

    def c4_maxpool():
        lay = tfg.layers.MaxPoolGraphSage(256, activation=tfg.relu)
        ms = _time(lambda: lay([x, ei, w1], cache=cache), steps=4, warmup=2)
        # the launch that dominates it, alone: the max reduce over the [N, 4 * ku = 512] per-node MLP rows
        # (as the layer lays them out: rows at plan.gather_friendly_ld — 544 floats apart, never a 2 KB power-of-two stride)
        width = int(lay.neighbor_mlp_kernel.shape[1])
        h = gemm_bias_act(x, lay.neighbor_mlp_kernel.detach(), act=L.ACT_RELU, out=P.gather_friendly_empty(n, width, dev))
        out = torch.empty((n, width), dtype=torch.float32, device=dev)
        ms_red = _time(lambda: segment_reduce(plan, h, L.MAX, out=out), steps=4, warmup=2)
        balg = b_alg(e, n, width, weighted=False)
        kname = segment_reduce(plan, h, L.MAX, out=out, describe=True)
        del h, out
        tl = tfg.layers.MaxPoolGraphSage(256, activation=tfg.relu)
        tl._maybe_build([x])
        tl.trainable(True)
        ms_t = _time(_fwd_bwd(tl, [x, ei, w1], cache), steps=3, warmup=2)
        return {"layer": "MaxPoolGraphSage(256, concat=True): per-node MLP {} -> {} (+ ReLU), max over in-edges at {} columns, "
                         "512 -> 128 and {} -> 128 projections".format(f, width, width, f),
                "forward_ms": ms, "fwd_bwd_ms": ms_t, "reduce_alone_kernel": kname, "reduce_alone_ms": ms_red,
                "reduce_columns": width, "row_stride_floats": P.gather_friendly_ld(width),
                "roofline": hbm_roof(balg, ms_red, kernel="the 512-column max reduce alone (5.2 GB table)"),
                "reduce_edges_per_s": e / (ms_red * 1e-3)}

    def width_sweep():
        rows = []
        wsum = torch.rand(e, device=dev) + 0.5
        sc = torch.rand(n, device=dev)
        for width in (128, 192, 256, 384, 512):
            xw = torch.randn(n, width, device=dev)
            ow = torch.empty_like(xw)
            ms = _time(lambda: segment_reduce(plan, xw, L.SUM, w_csr=wsum, self_coef=sc, out=ow), steps=5, warmup=2)
            balg = b_alg(e + n, n, width)
            rows.append({"F": width, "kernel": segment_reduce(plan, xw, L.SUM, w_csr=wsum, self_coef=sc, out=ow, describe=True),
                         "ms": ms, "roofline": hbm_roof(balg, ms),
                         "G_row_lines_per_s": (e + n) * (width * 4 / 128.0) / (ms * 1e-3) / 1e9})
            del xw, ow
        return {"what": "the headline launch (weighted sum + implicit self-loops) on the same graph at wider rows", "rows": rows}

    _entry(res, "C4_mean_sage_256_concat", c4_mean)
    _entry(res, "C4_maxpool_sage_256_concat", c4_maxpool)
    _entry(res, "C4_width_sweep", width_sweep)
    del w1

    # ---- dense x @ W beside torch.matmul (hipBLASLt), alternating
    def gemm_pair(m, k, nn, xin=None):
        a = xin if xin is not None else torch.randn(m, k, device=dev)
        b = L.as_f32(synthetic.glorot_uniform(k, nn))
        c = torch.empty((m, nn), dtype=torch.float32, device=dev)
        ours, lib = _time_ab(lambda: gemm_bias_act(a, b, out=c), lambda: torch.matmul(a, b, out=c), steps=10, warmup=3, rounds=3)
        return {"shape": "{} x {} -> {}".format(m, k, nn), "ms": ours, "torch_matmul_ms": lib, "ratio_ours_over_torch": ours / lib,
                "tflops_fp32": 2.0 * m * k * nn / (ours * 1e-3) / 1e12, "entry": "tfgx_gemm_bias_act_f32 (fp32 MFMA)",
                "roofline": max([roof("mfma", 2.0 * m * k * nn / (ours * 1e-3) / 1e12, 155.0, "TFLOP/s"),
                                 roof("hbm", 4.0 * (m * k + k * nn + m * nn) / (ours * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s",
                                      algorithmic_bytes=4 * (m * k + k * nn + m * nn))], key=lambda r_: r_["frac"])}

    _entry(res, "gemm_products_100_to_256", lambda: gemm_pair(n, f, 256, x if f == 100 else None))
    _entry(res, "gemm_arxiv_128_to_256", lambda: gemm_pair(170000, 128, 256))

    # ---- C2: 2-layer GCN 128 -> 256 -> 40 on the arxiv-shaped graph: forward and one full-batch training step
    def c2():
        na, ea, fa = synthetic.WORKLOADS["arxiv"]
        eia = L.as_i32(synthetic.synthetic_edge_stripe(na, ea, seed=seed))
        xa = L.as_f32(synthetic.synthetic_feature_rows(na, fa, seed=seed + 1))
        ca = {}
        g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)
        before = P.FUSED_STATS["launches"]
        fwd = lambda: g1([g0([xa, eia], cache=ca), eia], cache=ca)          # noqa: E731
        ms_f = _time(fwd, steps=20, warmup=5)
        fused = P.FUSED_STATS["launches"] > before
        g0.trainable(True)
        g1.trainable(True)
        opt = torch.optim.Adam(g0.parameters() + g1.parameters(), lr=1e-2)
        idx = torch.arange(0, na, 10, device=dev)
        labels = torch.randint(0, 40, (int(idx.shape[0]),), device=dev)

        def step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(fwd()[idx], labels).backward()
            opt.step()

        ms_s = _time(step, steps=10, warmup=3)
        ea_real = int(eia.shape[1])
        planA = CsrPlan.from_cache(eia, na, na, ca)
        balg0 = b_alg(ea_real + na, na, fa)           # layer 0 aggregates at the input width (128), layer 1 at 40
        # the layer-0 propagation alone (A_hat x at 128 columns, the launch the fused kernel's producers do the work of): its
        # 87 MB table is resident in the 256 MB Infinity Cache, so the roof is that cache's random-line rate ON THIS BOX
        # (512-byte aligned spans out of a table of the same size), not the HBM peak
        g0.trainable(False)
        from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj
        normedA = gcn_norm_adj(tfg.SparseMatrix(eia, None, [na, na]), cache={})
        outA = torch.empty_like(xa)
        ms_agg = _time(lambda: segment_reduce(normedA.plan, xa, L.SUM, w_csr=normedA.w_csr, self_coef=normedA.self_coef,
                                              out=outA), steps=30, warmup=5)
        lines = (ea_real + na) * (fa * 4 / 128.0)
        table_mib = na * fa * 4 / 2.0 ** 20
        probe = line_rate_probe(table_mib)
        peak = probe.get(512)
        return {"model": "GCN(256, relu) -> GCN(40) on N={} E={} F={} (demo/demo_gcn.py:18-32)".format(na, ea_real, fa),
                "forward_ms": ms_f, "train_step_ms": ms_s,
                "dominant_launch": "tfgx_aggregate_gemm_f32 -> agg_gemm_kernel<32, true> (A_hat x at 128 columns, x W fused)" if fused
                                   else "tfgx_segment_reduce_f32 + tfgx_gemm_bias_act_f32",
                "layer1_aggregation_kernel": segment_reduce(planA, torch.empty((na, 40), device=dev), L.SUM,
                                                            w_csr=torch.empty(ea_real, device=dev), describe=True),
                "layer0_aggregation_alone_ms": ms_agg,
                "roofline": roof("mall", lines / (ms_agg * 1e-3) / 1e9, peak, "G lines/s",
                                 kernel="layer-0 propagation alone: " + segment_reduce(
                                     normedA.plan, xa, L.SUM, w_csr=normedA.w_csr, self_coef=normedA.self_coef, out=outA, describe=True),
                                 peak_is="tools/line_rate_probe.cpp on this box in this run: random aligned 512-byte spans out of a "
                                         "{:.0f} MiB table (Infinity-Cache resident)".format(table_mib),
                                 probe_G_lines_per_s=probe, row_lines_per_launch=lines, algorithmic_bytes=balg0,
                                 algorithmic_GBps=balg0 / (ms_agg * 1e-3) / 1e9),
                "algorithmic_GBps_forward": (balg0 + b_alg(ea_real + na, na, 40)) / (ms_f * 1e-3) / 1e9}

    _entry(res, "C2_arxiv_gcn_2layer", c2)

    # ---- C3: demo/demo_gat.py:22 literal layer on the Reddit-shaped graph (attention_units = 8, d_head = 1) and SURVEY.md
    # 8d's heavy variant (attention_units = 64, d_head = 8): forward, forward + backward, attention alone
    reddit = {}

    def c3(A):
        from tf_geometric_amd.nn.conv import gat as G_
        nr, er, fr = synthetic.WORKLOADS["reddit"]
        if not reddit:
            reddit["ei"] = L.as_i32(synthetic.synthetic_edge_stripe(nr, er, seed=seed + 3))
            reddit["x"] = torch.randn(nr, fr, device=dev)
            reddit["cache"] = {}
        eir, xr, cr = reddit["ei"], reddit["x"], reddit["cache"]
        er_real = int(eir.shape[1])
        H, U = 8, 64
        lay = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
        ms_f = _time(lambda: lay([xr, eir], cache=cr), steps=8, warmup=2)
        planR = CsrPlan.from_cache(eir, nr, nr, cr)
        Q, K, V = (torch.randn(nr, A, device=dev), torch.randn(nr, A, device=dev), torch.randn(nr, U, device=dev))
        ms_att = _time(lambda: gat_attention(planR, Q, K, V, H), steps=8, warmup=2)
        tl = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
        tl._maybe_build([xr])
        tl.trainable(True)
        ms_t = _time(_fwd_bwd(tl, [xr, eir], cr), steps=4, warmup=2)
        e_agg = er_real + nr
        balg = e_agg * (4 * A + 4 * U + 4) + nr * 4 * (A + U) + 4 * (nr + 1)
        # what bounds it: every edge gathers one K row (4A bytes: 1 line at A = 8, 2 at A = 64) and one V row (256 bytes: 2
        # lines) out of a K | V source block that the block policy sized for the 4 MiB L2s — 128-byte lines served by L2
        kb = G_.source_block_count(planR, A, U)
        lines_per_edge = max(1, -(-4 * A // 128)) + 2
        lines = e_agg * lines_per_edge
        block_mib = nr * (A + U) * 4 / 2.0 ** 20 / max(kb, 1)
        probe = line_rate_probe(block_mib)
        return {"layer": "GAT(64, num_heads=8, attention_units={}) on N={} E={} F={}".format(A, nr, er_real, fr),
                "forward_ms": ms_f, "fwd_bwd_ms": ms_t, "fwd_bwd_over_forward": ms_t / ms_f,
                "backward": ("dQ per row from the sums the training forward accumulates (tfgx_gat_args.qgrad_t: one attention unit "
                             "per head), dK / dV in one source pass over destination blocks" if A == H else
                             "dQ in a destination pass over source blocks, dK / dV in a source pass over destination blocks"),
                "dominant_launch": "tfgx_gat_fused_f32 -> gat_fused_kernel (scores + online softmax + value sum), {} chained "
                                   "launches over source blocks of {:.1f} MiB".format(kb, block_mib),
                "attention_alone_ms": ms_att,
                "roofline": roof("l2", lines * 128 / (ms_att * 1e-3) / 1e9, L2_PEAK / 1e9, "GB/s",
                                 kernel="the attention alone ({} x gat_fused_kernel)".format(kb),
                                 achieved_is="128-byte lines gathered per launch sequence ({} per edge) x 128 B / time".format(lines_per_edge),
                                 lines_per_launch=lines, G_lines_per_s=lines / (ms_att * 1e-3) / 1e9,
                                 probe_G_lines_per_s_same_box=probe,
                                 probe_is="tools/line_rate_probe.cpp, random aligned spans out of a table of one source block's size",
                                 algorithmic_bytes=balg, algorithmic_GBps=balg / (ms_att * 1e-3) / 1e9),
                "edges_per_s_layer": er_real / (ms_f * 1e-3)}

    _entry(res, "C3_reddit_gat_H8_A8", lambda: c3(8))
    _entry(res, "C3_reddit_gat_H8_A64", lambda: c3(64))
    reddit.clear()
    torch.cuda.empty_cache()

    # ---- C5: ONE of the 8 destination shards of the papers100M shape (13.9 M destination rows, 200 M in-edges, sources anywhere
    # among 111 M nodes: the 56.8 GB source table — own rows + halo — resident), the weighted segment-sum at F = 128
    def c5():
        free, _ = torch.cuda.mem_get_info()
        n_src, n_dst, e5, f5 = 111000000, 13875000, 200000000, 128
        if free < 75 * 2 ** 30:
            return {"skipped": "needs ~70 GB of free HBM, {:.0f} GB free".format(free / 2 ** 30)}
        g5 = torch.Generator(device=dev)
        g5.manual_seed(seed + 12)
        ei5 = torch.stack([torch.randint(0, n_dst, (e5,), generator=g5, device=dev, dtype=torch.int32),
                           torch.randint(0, n_src, (e5,), generator=g5, device=dev, dtype=torch.int32)])
        w5 = torch.rand(e5, generator=g5, device=dev) + 0.5
        x5 = torch.empty(n_src, f5, device=dev)
        for i in range(0, n_src, 8000000):                      # generated in slabs: no 57 GB temporary
            x5[i:i + 8000000].normal_(generator=g5)
        plan5 = CsrPlan.build(ei5, n_dst, n_src)
        w5c = plan5.edge_attr_to_csr(w5)
        del ei5, w5
        out5 = torch.empty((n_dst, f5), dtype=torch.float32, device=dev)
        ms5 = _time(lambda: segment_reduce(plan5, x5, L.SUM, w_csr=w5c, out=out5), steps=5, warmup=2)
        balg = e5 * (4 * f5 + 8) + n_dst * 4 * f5 + 4 * (n_dst + 1)
        kname = segment_reduce(plan5, x5, L.SUM, w_csr=w5c, out=out5, describe=True)
        del x5, out5, plan5, w5c
        torch.cuda.empty_cache()
        return {"shard": "1 of 8 destination shards of the papers100M shape: {} destination rows, {} in-edges, {} source rows x "
                         "{} floats = {:.1f} GB table".format(n_dst, e5, n_src, f5, n_src * f5 * 4 / 1e9),
                "aggregation_ms": ms5, "edges_per_s": e5 / (ms5 * 1e-3),
                "roofline": hbm_roof(balg, ms5, kernel=kname)}

    _entry(res, "C5_papers100M_one_shard_of_8", c5)
    P.AUTO_STATIC_LAYOUT = auto
    res["wall_s"] = time.perf_counter() - t_start
    return res


def extras(tfg, L, synthetic, x, ei, n, e, f, cache):
    """Whole-layer timings on the same graph (ms): GEMM + aggregation through the layer API.  `*_static_ms` keys are
    measured after tfg.prepare_static_features(x, ...) (explicit opt-in); everything else reads x as
    it is."""
    res = {}
    from tf_geometric_amd import plan as P
    P.AUTO_STATIC_LAYOUT = False      # the plain keys below time x AS IT IS; the *_static_* keys are measured after the explicit
                                      # opt-in — the state the automatic promotion reaches on a layer's second call
    w1 = torch.ones(e, dtype=torch.float32, device=x.device)
    gcn = tfg.layers.GCN(256, activation=tfg.relu)
    sage = tfg.layers.MeanGraphSage(256)
    res["gcn_layer_F{}_to_256_ms".format(f)] = _time(lambda: gcn([x, ei], cache=cache))
    res["mean_sage_layer_units256_ms"] = _time(lambda: sage([x, ei, w1], cache=cache))
    mp = tfg.layers.MaxPoolGraphSage(64)
    res["maxpool_sage_layer_units64_ms"] = _time(lambda: mp([x, ei, w1], cache=cache))
    gat = tfg.layers.GAT(64, attention_units=8, num_heads=8, activation=tfg.relu)
    res["gat_layer_H8_A8_U64_ms"] = _time(lambda: gat([x, ei], cache=cache))
    # 2-layer GCN (F -> 256 -> 40, BASELINE configs[1] model): eager launches vs one hipGraph replay; the model closes
    # over its static input features (the hipGraph captures their address either way)
    g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)

    def two_layer():
        return g1([g0([x, ei], cache=cache), ei], cache=cache)

    cap = tfg.CapturedForward(two_layer)
    det = {}
    res["gcn_2layer_eager_ms"], res["gcn_2layer_hipgraph_ms"] = _time_ab(two_layer, lambda: cap.graph.replay(), detail=det)
    res["gcn_2layer_eager_vs_hipgraph_rounds_ms"] = det["rounds_ms"]
    # the two numbers above are THROUGHPUT (steps queued back to back: the host runs ahead, launch cost is hidden, a
    # graph replay cannot win); what a hipGraph removes is per-step host LATENCY — one forward, then wait for it:
    res["gcn_2layer_eager_latency_ms"] = _latency(two_layer)
    res["gcn_2layer_hipgraph_latency_ms"] = _latency(lambda: cap.graph.replay())
    info = tfg.prepare_static_features(x, ei, cache)
    res["static_layout_bytes"] = int(info["bytes"])
    res["gcn_layer_F{}_to_256_static_ms".format(f)] = _time(lambda: gcn([x, ei], cache=cache))
    res["mean_sage_layer_units256_static_ms"] = _time(lambda: sage([x, ei, w1], cache=cache))
    cap2 = tfg.CapturedForward(two_layer)                 # prepared BEFORE capture: the replay runs the static layout
    det = {}
    res["gcn_2layer_eager_static_ms"], res["gcn_2layer_hipgraph_static_ms"] = _time_ab(two_layer,
                                                                                       lambda: cap2.graph.replay(), detail=det)
    res["gcn_2layer_eager_vs_hipgraph_static_rounds_ms"] = det["rounds_ms"]
    res["gcn_2layer_eager_static_latency_ms"] = _latency(two_layer)
    res["gcn_2layer_hipgraph_static_latency_ms"] = _latency(lambda: cap2.graph.replay())
    tfg.release_static_features(cache)
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

    torch.cuda.reset_peak_memory_stats()
    res["gcn_2layer_train_step_ms"] = _time(full_step, steps=5, warmup=2)
    res["gcn_2layer_train_step_peak_GB"] = torch.cuda.max_memory_allocated() / 1e9      # graph + plans + activations + workspaces
    # the same step replayed from ONE hipGraph (tfg.CapturedTrainStep: zero-grad, forward, loss, backward, Adam update),
    # measured alternately with the eager loop on the same capturable optimizer; a step is launch-bound on small graphs
    # (arxiv shape), kernel-bound at products shape
    opt_c = torch.optim.Adam(t0l.parameters() + t1l.parameters(), lr=1e-2, capturable=True)

    def loss_c():
        return torch.nn.functional.cross_entropy(t1l([t0l([x, ei], cache=cache), ei], cache=cache)[idx], labels)

    def full_step_c():
        opt_c.zero_grad(set_to_none=True)
        loss_c().backward()
        opt_c.step()

    cap_step = tfg.CapturedTrainStep(loss_c, opt_c)
    det = {}
    res["gcn_2layer_train_step_eager_capturable_ms"], res["gcn_2layer_train_step_hipgraph_ms"] = _time_ab(
        full_step_c, cap_step, steps=5, warmup=2, rounds=3, detail=det)
    res["gcn_2layer_train_step_eager_vs_hipgraph_rounds_ms"] = det["rounds_ms"]
    res["gcn_2layer_train_step_eager_latency_ms"] = _latency(full_step_c, steps=10)
    res["gcn_2layer_train_step_hipgraph_latency_ms"] = _latency(cap_step, steps=10)
    del cap_step
    # BASELINE configs[3]: GraphSAGE mean / max-pool aggregators (units 256, concat, as demo/demo_graph_sage.py:29-30):
    # one full-batch training step of a 2-layer mean model, and forward + backward of one max-pool layer
    s0, s1 = tfg.layers.MeanGraphSage(256, activation=tfg.relu), tfg.layers.MeanGraphSage(40, activation=None)
    s0.trainable(True)
    s1.trainable(True)
    with torch.no_grad():
        s1([s0([x, ei, w1], cache=cache), ei, w1], cache=cache)
    opt2 = torch.optim.Adam(s0.parameters() + s1.parameters(), lr=1e-2)

    def sage_step():
        opt2.zero_grad(set_to_none=True)
        logits = s1([s0([x, ei, w1], cache=cache), ei, w1], cache=cache)
        torch.nn.functional.cross_entropy(logits[idx], labels).backward()
        opt2.step()

    res["mean_sage_2layer_train_step_ms"] = _time(sage_step, steps=4, warmup=2)
    # the same two training steps with the input features declared static AND the layer-0 aggregation memoised
    # (prepare_static_features(..., cache_aggregation=True)): A_hat @ x / mean(x[col]) never change while x and the edge
    # weights do not, so layer 0's aggregation runs once, not once per step — reported beside, never instead
    tfg.prepare_static_features(x, ei, cache, cache_aggregation=True)
    res["gcn_2layer_train_step_static_agg_memo_ms"] = _time(full_step, steps=5, warmup=2)
    res["mean_sage_2layer_train_step_static_agg_memo_ms"] = _time(sage_step, steps=4, warmup=2)
    tfg.release_static_features(cache)
    mpt = tfg.layers.MaxPoolGraphSage(64)
    mpt._maybe_build([x])
    mpt.trainable(True)

    def maxpool_fwd_bwd():
        for p_ in mpt.parameters():
            p_.grad = None
        mpt([x, ei, w1], cache=cache).sum().backward()

    res["maxpool_sage_layer_units64_fwd_bwd_ms"] = _time(maxpool_fwd_bwd, steps=4, warmup=2)
    from tf_geometric_amd.plan import gemm_bias_act
    k = L.as_f32(synthetic.glorot_uniform(f, 256))
    ms = _time(lambda: gemm_bias_act(x, k))
    res["gemm_{}x{}x256_ms".format(n, f)] = ms
    res["gemm_tflops"] = 2.0 * n * f * 256 / (ms * 1e-3) / 1e12
    P.AUTO_STATIC_LAYOUT = True
    return res


if __name__ == "__main__":
    main()
