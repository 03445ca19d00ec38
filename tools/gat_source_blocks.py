# coding=utf-8
"""Experiment (round 5): fused GAT attention at the Reddit shape with the edges of every destination row partitioned by
SOURCE BLOCK — pass k gathers K / V rows of block k only, so the gathered table of a pass (N / KB rows x 288 bytes) fits the
4 MB L2 of every XCD instead of living in the Infinity Cache; raw online-softmax states of the KB passes are merged by
tfgx_gat_merge_passes_f32.  Prints one JSON line per KB with the time of the passes + merge against the one-pass kernel and
the max abs difference of the outputs (the merge re-associates the softmax sums).

    python tools/gat_source_blocks.py [A=8] [KB list, e.g. 8,16,32]"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import synthetic, _lib as L, plan as P              # noqa: E402
from tf_geometric_amd.nn.conv.gat import gat_attention, gat_args           # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blocks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8, 16, 32, 64]
H, U = 8, 64
n, e, _ = synthetic.WORKLOADS["reddit"]
dev = torch.device("cuda")
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
E = int(ei.shape[1])
plan = P.CsrPlan.build(ei, n, n)
g = torch.Generator(device=dev)
g.manual_seed(5)
Q = torch.randn(n, A, generator=g, device=dev)
K = torch.randn(n, A, generator=g, device=dev)
V = torch.randn(n, U, generator=g, device=dev)
lib = L.require_gpu()


def timeit(fn, steps=8, warmup=2):
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


ref = gat_attention(plan, Q, K, V, H)
ms_ref = timeit(lambda: gat_attention(plan, Q, K, V, H))
print(json.dumps({"what": "one pass (shipped)", "A": A, "ms": ms_ref}), flush=True)
deg = plan.row_ptr[1:] - plan.row_ptr[:-1]
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), deg.long())
for KB in blocks:
    blk = -(-n // KB)
    key = rows * KB + (plan.col.long() // blk)
    order = torch.argsort(key, stable=True)
    col_k = plan.col[order].contiguous()
    rpk = torch.zeros(n * KB + 1, dtype=torch.int32, device=dev)
    rpk[1:] = torch.cumsum(torch.bincount(key, minlength=n * KB), 0).to(torch.int32)
    s_acc = torch.empty((KB, n, U), dtype=torch.float32, device=dev)
    s_ml = torch.empty((KB, n, 2 * H), dtype=torch.float32, device=dev)
    out = torch.empty((n, U), dtype=torch.float32, device=dev)

    def run():
        for k in range(KB):
            a, _, keep = gat_args(Q, K, V, H, n, col_k, add_self_loop=False, out=s_acc[k])
            a.row_begin, a.row_end, a.rp_stride = rpk[k:].data_ptr(), rpk[k + 1:].data_ptr(), KB
            a.state_acc, a.state_ml = s_acc[k].data_ptr(), s_ml[k].data_ptr()
            L.check(lib.tfgx_gat_fused_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_gat_fused_f32")
        a, _, keep = gat_args(Q, K, V, H, n, col_k, add_self_loop=True, out=out)
        L.check(lib.tfgx_gat_merge_passes_f32(ctypes.byref(a), L.ptr(s_acc), L.ptr(s_ml), KB, L.stream_ptr()),
                "tfgx_gat_merge_passes_f32")

    run()
    torch.cuda.synchronize()
    err = float((out - ref).abs().max().item())
    ms = timeit(run)

    def passes_only():
        for k in range(KB):
            a, _, keep = gat_args(Q, K, V, H, n, col_k, add_self_loop=False, out=s_acc[k])
            a.row_begin, a.row_end, a.rp_stride = rpk[k:].data_ptr(), rpk[k + 1:].data_ptr(), KB
            a.state_acc, a.state_ml = s_acc[k].data_ptr(), s_ml[k].data_ptr()
            L.check(lib.tfgx_gat_fused_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_gat_fused_f32")

    ms_p = timeit(passes_only)
    print(json.dumps({"what": "source blocks", "A": A, "KB": KB, "block_rows": blk, "block_table_MB": blk * (A + U) * 4 / 1e6,
                      "ms_passes_plus_merge": ms, "ms_passes": ms_p, "speedup_vs_one_pass": ms_ref / ms,
                      "max_abs_diff_vs_one_pass": err, "state_bytes": int(s_acc.numel() + s_ml.numel()) * 4}), flush=True)
    del s_acc, s_ml, col_k, rpk, order, key
