# coding=utf-8
"""Kernel-level sweeps on one GPU (not the contract bench): segment-reduce time vs feature width / reducer / graph
skew, GEMM TFLOP/s, fused GAT.  Prints one JSON object per line.  usage: python tools/bench_sweep.py [--quick]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                      # noqa: E402
from tf_geometric_amd import _lib as L, synthetic   # noqa: E402
from tf_geometric_amd.plan import CsrPlan, segment_reduce, gemm_bias_act   # noqa: E402
from tf_geometric_amd.nn.conv.gat import gat_attention                      # noqa: E402


def timeit(fn, steps=10, warmup=3):
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


def rmat_edges(scale, num_edges, seed=0, a=0.57, b=0.19, c=0.19):
    """R-MAT (a,b,c,d) = (.57,.19,.19,.05) on 2^scale nodes, generated on the GPU; both directions emitted."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    half = num_edges // 2
    src = torch.zeros(half, dtype=torch.int64, device="cuda")
    dst = torch.zeros(half, dtype=torch.int64, device="cuda")
    for _ in range(scale):
        r = torch.rand(half, generator=g, device="cuda")
        sbit = (r >= a + b).to(torch.int64)
        dbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
        src = src * 2 + sbit
        dst = dst * 2 + dbit
    keep = src != dst
    src, dst = src[keep], dst[keep]
    return torch.stack([torch.cat([src, dst]), torch.cat([dst, src])]).to(torch.int32)


def main():
    quick = "--quick" in sys.argv
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    want = lambda name: (not only) or (name in only[0].split(","))
    torch.cuda.set_device(0)
    n, e = (2400000, 123000000) if not quick else (300000, 15000000)
    if want("reddit"):
        reddit_section(quick)
    if want("papers_shard"):
        papers_shard_section(quick)
    if only and "papers_full" in only[0].split(","):        # explicit only: ~140 GB of HBM
        papers_full_section(quick)
    if want("gemm"):
        gemm_section(n)
    if only and not any(want(s) for s in ("widths", "split", "rmat", "gat", "backward")):
        return
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
    E = int(ei.shape[1])
    plan = CsrPlan.build(ei, n, n)
    w = torch.rand(E, device="cuda") + 0.5
    for f in ([8, 16, 32, 40, 64, 96, 100, 104, 128, 192, 256] if want("widths") else []):
        x = torch.randn(n, f, device="cuda")
        out = torch.empty_like(x)
        for name, op, ww in [("sum_w", L.SUM, w), ("sum", L.SUM, None), ("max", L.MAX, None)]:
            if name != "sum_w" and f not in (100, 128):
                continue
            ms = timeit(lambda: segment_reduce(plan, x, op, w_csr=ww, out=out))
            balg = E * (4 * f + 4 + (4 if ww is not None else 0)) + n * 4 * f + 4 * (n + 1)
            print(json.dumps({"kind": "segment_reduce", "graph": "uniform", "N": n, "E": E, "F": f, "op": name,
                              "ms": ms, "GBps_alg": balg / ms / 1e6, "frac_of_8TBps": balg / ms / 1e6 / 8000,
                              "Gedges_per_s": E / ms / 1e6}), flush=True)
        del x, out
    if want("split"):
        from tf_geometric_amd.plan import SplitRows
        for f in [100, 104, 72]:
            x = torch.randn(n, f, device="cuda")
            out = torch.empty_like(x)
            sp = SplitRows.from_dense(x)
            balg = E * (4 * f + 8) + n * 4 * f + 4 * (n + 1)
            ms_d = timeit(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out))
            ms_k = timeit(lambda: segment_reduce(plan, sp, L.SUM, w_csr=w, out=out))
            ms_c = timeit(lambda: SplitRows.from_dense(x, out=sp))
            ms_b = timeit(lambda: segment_reduce(plan, SplitRows.from_dense(x, out=sp), L.SUM, w_csr=w, out=out))
            t0 = time.perf_counter()
            sp.with_edge_tail(plan)
            torch.cuda.synchronize()
            et_build_ms = (time.perf_counter() - t0) * 1e3
            ms_e = timeit(lambda: segment_reduce(plan, sp, L.SUM, w_csr=w, out=out))
            assert torch.equal(out, segment_reduce(plan, x, L.SUM, w_csr=w))
            sp.edge_tail = None
            print(json.dumps({"kind": "split_rows", "F": f, "dense_ms": ms_d, "split_kernel_ms": ms_k,
                              "edge_tail_kernel_ms": ms_e, "edge_tail_build_ms": et_build_ms,
                              "frac_edge_tail_kernel": balg / ms_e / 8e9,
                              "split_convert_ms": ms_c, "convert_plus_kernel_ms": ms_b,
                              "frac_dense": balg / ms_d / 8e9, "frac_split_kernel": balg / ms_k / 8e9,
                              "frac_convert_plus_kernel": balg / ms_b / 8e9}), flush=True)
            del x, out, sp
    if want("rmat"):
        rmat_section(quick, e)
    if want("gat"):
        gat_section(plan, n, E)
    if want("backward"):
        backward_section(plan, n, E, w)


def rmat_section(quick, e):
    # skewed graph (R-MAT): hub rows take the chunked path
    scale = 21 if not quick else 18
    ei_r = rmat_edges(scale, e // 2 if not quick else e // 4, seed=1)
    nr, Er = 1 << scale, int(ei_r.shape[1])
    plan_r = CsrPlan.build(ei_r, nr, nr)
    deg = plan_r.in_degree()
    hub = plan_r.hub_info()
    x = torch.randn(nr, 100, device="cuda")
    out = torch.empty_like(x)
    wr = plan_r.edge_attr_to_csr(torch.rand(Er, device="cuda") + 0.5)     # weights in CSR order
    ms = timeit(lambda: segment_reduce(plan_r, x, L.SUM, w_csr=wr, out=out))
    balg = Er * (400 + 8) + nr * 400 + 4 * (nr + 1)
    print(json.dumps({"kind": "segment_reduce", "graph": "rmat(.57,.19,.19,.05)", "N": nr, "E": Er, "F": 100,
                      "max_in_degree": int(deg.max().item()), "hub_rows": 0 if hub is None else int(hub[0].shape[0]),
                      "hub_chunks": 0 if hub is None else int(hub[2].shape[0]), "ms": ms,
                      "GBps_alg": balg / ms / 1e6, "frac_of_8TBps": balg / ms / 1e6 / 8000,
                      "Gedges_per_s": Er / ms / 1e6}), flush=True)
    # hub threshold / chunk size sweep + degree histogram
    import tf_geometric_amd.plan as P
    hist = torch.histc(torch.log2(deg.float().clamp(min=1)), bins=18, min=0, max=18).long().tolist()
    edges_by_bin = []
    lg = torch.log2(deg.float().clamp(min=1)).floor().long().clamp(max=17)
    eb = torch.zeros(18, dtype=torch.int64, device="cuda").index_add_(0, lg, deg.long())
    print(json.dumps({"kind": "rmat_degree_hist_log2", "rows": hist, "edges": eb.tolist()}), flush=True)
    for thr, chunk in [(128, 128), (256, 256), (512, 256), (512, 512), (1024, 512), (2048, 1024), (8192, 1024), (1 << 30, 1024)]:
        P.HUB_THRESHOLD, P.HUB_CHUNK = thr, chunk
        plan_r._hub = None
        ms = timeit(lambda: segment_reduce(plan_r, x, L.SUM, w_csr=wr, out=out))
        hub = plan_r.hub_info()
        print(json.dumps({"kind": "rmat_hub_sweep", "threshold": thr, "chunk": chunk, "ms": ms,
                          "hub_rows": 0 if hub is None else int(hub[0].shape[0]),
                          "hub_chunks": 0 if hub is None else int(hub[2].shape[0]), "Gedges_per_s": Er / ms / 1e6}),
              flush=True)
    P.HUB_THRESHOLD, P.HUB_CHUNK = None, None
    plan_r._hub = None
    hub = plan_r.hub_info()
    segment_reduce(plan_r, x, L.SUM, w_csr=wr, out=out)
    # parity of the hub path against a float64 torch reference on the hub rows
    if hub is not None:
        rows = hub[0][:4].long()
        got = out[rows].double().cpu()
        ref = torch.zeros(len(rows), 100, dtype=torch.float64)
        rp = plan_r.row_ptr.long()
        for i, r in enumerate(rows.tolist()):
            s, t = int(rp[r]), int(rp[r + 1])
            ref[i] = (x[plan_r.col[s:t].long()].double() * wr[s:t, None].double()).sum(0).cpu()
        print(json.dumps({"kind": "hub_parity", "max_abs_err": float((got - ref).abs().max()),
                          "max_abs_ref": float(ref.abs().max())}), flush=True)
    del x, out


def gemm_section(n):
    for (m, k, nn) in [(n, 100, 256), (n, 100, 128), (n, 128, 256), (n, 100, 64), (n, 256, 128), (n, 100, 16),
                       (n, 256, 256), (170000, 128, 256), (170000, 256, 40), (233000, 602, 64), (2708 * 64, 1433, 16),
                       (233000, 602, 16), (170000, 1433, 256), (100000, 301, 40), (n, 256, 40)]:
        a = torch.randn(m, k, device="cuda")
        b = torch.randn(k, nn, device="cuda") * 0.1
        c = torch.empty(m, nn, device="cuda")
        ms = timeit(lambda: gemm_bias_act(a, b, out=c))
        ms_t = timeit(lambda: torch.matmul(a, b, out=c))
        print(json.dumps({"kind": "gemm", "M": m, "K": k, "N": nn, "ms": ms, "TFLOPs": 2.0 * m * k * nn / ms / 1e9,
                          "GBps": 4.0 * (m * k + m * nn) / ms / 1e6, "torch_matmul_ms": ms_t}), flush=True)
        del a, b, c


def gat_section(plan, n, E):
    # fused GAT attention (demo-literal H=8, A=8, U=64 and the heavy A=64 variant)
    for (H, A, U) in [(8, 8, 64), (8, 64, 64), (1, 8, 64)]:
        Q = torch.randn(n, A, device="cuda")
        K = torch.randn(n, A, device="cuda")
        V = torch.randn(n, U, device="cuda")
        ms = timeit(lambda: gat_attention(plan, Q, K, V, H))
        Eagg = E + n
        balg = Eagg * (4 * A + 4 * U + 4) + n * 4 * (A + U) + 4 * (n + 1)
        print(json.dumps({"kind": "gat_fused", "H": H, "A": A, "U": U, "ms": ms, "GBps_alg": balg / ms / 1e6,
                          "frac_of_8TBps": balg / ms / 1e6 / 8000, "Gedges_per_s": Eagg / ms / 1e6}), flush=True)


def reddit_section(quick):
    """BASELINE.json configs[2]: multi-head GAT on the Reddit-shaped graph (N=233k, E=114M, F=602, avg in-degree 489).
    demo/demo_gat.py:22 literal layer GAT(64, num_heads=8, attention_units=8), the heavy attention_units=64 variant,
    and the attention kernel alone."""
    import tf_geometric_amd as tfg
    n, e, f = synthetic.WORKLOADS["reddit"] if not quick else (233000, 11400000, 602)
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=3))
    E = int(ei.shape[1])
    plan = CsrPlan.build(ei, n, n)
    x = torch.randn(n, f, device="cuda")
    cache = {"tfgx_csr_plan": plan}
    for (H, A, U) in [(8, 8, 64), (8, 64, 64)]:
        layer = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
        ms_layer = timeit(lambda: layer([x, ei], cache=cache))
        Q = torch.randn(n, A, device="cuda")
        K = torch.randn(n, A, device="cuda")
        V = torch.randn(n, U, device="cuda")
        ms_att = timeit(lambda: gat_attention(plan, Q, K, V, H))
        Eagg = E + n
        balg = Eagg * (4 * A + 4 * U + 4) + n * 4 * (A + U) + 4 * (n + 1)
        # one training step of the layer: forward + backward through the fused attention kernels and the MFMA GEMMs
        tl = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
        tl._maybe_build([x])
        tl.trainable(True)

        def fwd_bwd():
            for p_ in tl.parameters():
                p_.grad = None
            tl([x, ei], cache=cache).sum().backward()
        ms_train = timeit(fwd_bwd, steps=4, warmup=2)
        print(json.dumps({"kind": "reddit_gat", "N": n, "E": E, "F": f, "H": H, "A": A, "U": U, "layer_ms": ms_layer,
                          "layer_fwd_bwd_ms": ms_train,
                          "attention_ms": ms_att, "attention_GBps_alg": balg / ms_att / 1e6,
                          "attention_frac_of_8TBps": balg / ms_att / 1e6 / 8000,
                          "Gedges_per_s_layer": E / ms_layer / 1e6}), flush=True)
    # the demo's whole model (demo/demo_gat.py:22-23): GAT(64, 8 heads, attention_units 8) -> GAT(41, 1 head,
    # attention_units 1) on the 41 Reddit classes; the second layer's odd value width runs zero-padded to 44 columns
    g0 = tfg.layers.GAT(64, attention_units=8, num_heads=8, activation=tfg.relu)
    g1 = tfg.layers.GAT(41, attention_units=1, num_heads=1)
    ms_model = timeit(lambda: g1([g0([x, ei], cache=cache), ei], cache=cache))
    h0 = g0([x, ei], cache=cache)
    ms_l1 = timeit(lambda: g1([h0, ei], cache=cache))
    g0.trainable(True)
    g1.trainable(True)

    def model_step():
        for p_ in g0.parameters() + g1.parameters():
            p_.grad = None
        g1([g0([x, ei], cache=cache), ei], cache=cache).sum().backward()
    ms_model_train = timeit(model_step, steps=3, warmup=2)
    print(json.dumps({"kind": "reddit_gat_model", "what": "GAT(64, H8, A8) -> GAT(41, H1, A1), demo/demo_gat.py:22-23",
                      "forward_ms": ms_model, "second_layer_ms": ms_l1, "fwd_bwd_ms": ms_model_train}), flush=True)
    del x, plan, ei


def papers_shard_section(quick):
    """BASELINE.json configs[4] seen from ONE of its 8 GPUs: ogbn-papers100M-shaped GCN (N = 111 M, E = 1.6 G, F = 128),
    destination rows [0, N/8) with their E/8 in-edges whose sources are spread over all N nodes — i.e. the rectangular
    aggregation a shard runs once its halo rows are resident (table = 111 M x 128 fp32 = 56.8 GB of the 288 GB)."""
    n_src = 111000000 if not quick else 11100000
    n_dst = n_src // 8
    E = 200000000 if not quick else 20000000
    f = 128
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    row = torch.randint(0, n_dst, (E,), generator=g, device="cuda", dtype=torch.int32)
    col = torch.randint(0, n_src, (E,), generator=g, device="cuda", dtype=torch.int32)
    plan = CsrPlan.build(torch.stack([row, col]), n_dst, n_src)
    del row, col
    x = torch.empty(n_src, f, device="cuda")
    for i in range(0, n_src, 1 << 24):                     # fill in slabs: randn of the whole table would double it
        x[i:i + (1 << 24)].normal_(generator=g)
    w = torch.rand(E, generator=g, device="cuda") + 0.5
    out = torch.empty(n_dst, f, device="cuda")
    ms = timeit(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out))
    balg = E * (4 * f + 8) + n_dst * 4 * f + 4 * (n_dst + 1)
    # spot parity on a few rows against float64
    rp = plan.row_ptr.long()
    rows = [0, n_dst // 3, n_dst - 1]
    err = 0.0
    for r in rows:
        s, t = int(rp[r]), int(rp[r + 1])
        ref = (x[plan.col[s:t].long()].double() * w[s:t, None].double()).sum(0)
        err = max(err, float((out[r].double() - ref).abs().max()))
    print(json.dumps({"kind": "papers100M_shard", "n_dst": n_dst, "n_src": n_src, "E": E, "F": f,
                      "table_GB": n_src * f * 4 / 1e9, "ms": ms, "GBps_alg": balg / ms / 1e6,
                      "frac_of_8TBps": balg / ms / 1e6 / 8000, "Gedges_per_s": E / ms / 1e6,
                      "spot_max_abs_err": err, "mem_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    del x, out, plan, w


def papers_full_section(quick):
    """BASELINE.json configs[4] on ONE GPU: the whole ogbn-papers100M-shaped graph (N = 111 M, E = 1.6 G, F = 128) —
    edge list, plan, features and output resident together (~140 GB of the 288 GB).  The HBM-capacity stress the
    config names, without the 8-way split: plan build time, one weighted segment-sum pass, spot parity."""
    n = 111000000 if not quick else 11100000
    E = 1600000000 if not quick else 160000000
    f = 128
    g = torch.Generator(device="cuda")
    g.manual_seed(12)
    row = torch.randint(0, n, (E,), generator=g, device="cuda", dtype=torch.int32)
    col = torch.randint(0, n, (E,), generator=g, device="cuda", dtype=torch.int32)
    ei = torch.stack([row, col])
    del row, col
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = CsrPlan.build(ei, n, n)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    plan._edge_index = None
    del ei
    torch.cuda.empty_cache()
    x = torch.empty(n, f, device="cuda")
    for i in range(0, n, 1 << 24):
        x[i:i + (1 << 24)].normal_(generator=g)
    w = torch.rand(E, generator=g, device="cuda") + 0.5
    out = torch.empty(n, f, device="cuda")
    ms = timeit(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out), steps=5, warmup=2)
    balg = E * (4 * f + 8) + n * 4 * f + 4 * (n + 1)
    rp = plan.row_ptr.long()
    err = 0.0
    for r in (0, n // 2, n - 1):
        s, t = int(rp[r]), int(rp[r + 1])
        ref = (x[plan.col[s:t].long()].double() * w[s:t, None].double()).sum(0)
        err = max(err, float((out[r].double() - ref).abs().max()))
    assert int(rp[-1]) == E
    print(json.dumps({"kind": "papers100M_full_one_gpu", "N": n, "E": E, "F": f, "plan_build_s": plan_s, "ms": ms,
                      "GBps_alg": balg / ms / 1e6, "frac_of_8TBps": balg / ms / 1e6 / 8000,
                      "Gedges_per_s": E / ms / 1e6, "spot_max_abs_err": err,
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
    del x, out, plan, w


def wr_csr(plan, w):
    return plan.edge_attr_to_csr(w)




def backward_section(plan, n, E, w):
    """Backward kernels at products shape: transposed aggregation (d/dx), SDDMM (d/dw), max gradient, GAT gradient."""
    import tf_geometric_amd.autograd as AG
    w_csr = plan.edge_attr_to_csr(w)
    for f in [100]:
        x = torch.randn(n, f, device="cuda", requires_grad=True)
        g = torch.randn(n, f, device="cuda")
        for name, op, ww, det in [("sum_w", L.SUM, w_csr, "mask"),
                                  ("max (mask: winner bit masks, 1 gather per edge, deterministic)", L.MAX, None, "mask"),
                                  ("max (push: arg positions + atomics)", L.MAX, None, "push"),
                                  ("max (pull: bit-reproducible, 2 gathers per edge)", L.MAX, None, "pull"),
                                  ("max weighted (mask)", L.MAX, w_csr, "mask")]:
            AG.MAX_GRADIENT_MODE = det
            fwd = timeit(lambda: AG.aggregate(plan, x, op, ww), steps=5, warmup=2)
            out = AG.aggregate(plan, x, op, ww)
            ms = timeit(lambda: torch.autograd.grad(out, x, g, retain_graph=True), steps=5, warmup=2)
            AG.MAX_GRADIENT_MODE = "mask"
            print(json.dumps({"kind": "backward", "what": "d/dx aggregate " + name, "F": f, "ms": ms,
                              "training_forward_ms": fwd}), flush=True)
        from tf_geometric_amd.plan import gemm_tn, gemm_bias_act, transpose
        for (ka, nn_) in [(100, 256), (256, 40), (128, 128)]:
            xa = torch.randn(n, ka, device="cuda")
            gg = torch.randn(n, nn_, device="cuda")
            kk = torch.randn(ka, nn_, device="cuda")
            ms_tn = timeit(lambda: gemm_tn(xa, gg, want_bias=True), steps=5, warmup=2)
            ms_ref = timeit(lambda: (xa.t() @ gg, gg.sum(0)), steps=5, warmup=2)
            ms_dx = timeit(lambda: gemm_bias_act(gg, transpose(kk)), steps=5, warmup=2)
            ms_dx_ref = timeit(lambda: gg @ kk.t(), steps=5, warmup=2)
            print(json.dumps({"kind": "backward", "what": "dense layer gradients", "M": n, "K": ka, "N": nn_,
                              "dW_db_gemm_tn_ms": ms_tn, "dW_db_torch_ms": ms_ref,
                              "dW_TFLOPs": 2.0 * n * ka * nn_ / (ms_tn * 1e-3) / 1e12,
                              "dx_mfma_ms": ms_dx, "dx_torch_ms": ms_dx_ref}), flush=True)
        wreq = w_csr.clone().requires_grad_(True)
        out = AG.aggregate(plan, x.detach(), L.SUM, wreq)
        ms = timeit(lambda: torch.autograd.grad(out, wreq, g, retain_graph=True), steps=5, warmup=2)
        print(json.dumps({"kind": "backward", "what": "d/dw (sddmm)", "F": f, "ms": ms}), flush=True)
    for (H, A, U) in [(8, 8, 64)]:
        Q = torch.randn(n, A, device="cuda", requires_grad=True)
        K = torch.randn(n, A, device="cuda", requires_grad=True)
        V = torch.randn(n, U, device="cuda", requires_grad=True)
        g = torch.randn(n, U, device="cuda")
        out = AG.gat_attention(plan, Q, K, V, H)
        ms = timeit(lambda: torch.autograd.grad(out, (Q, K, V), g, retain_graph=True), steps=3, warmup=1)
        print(json.dumps({"kind": "backward", "what": "GAT dQ,dK,dV", "H": H, "A": A, "U": U, "ms": ms}), flush=True)


if __name__ == "__main__":
    main()
