# coding=utf-8
"""Seeded random sweeps through the C ABI against the float64 oracle: shapes, widths (every vector / lane-group /
chunk dispatch), reducers, optional operands and ragged graphs (empty rows, duplicate edges, self-loops, one very long
row) drawn at random — the combinations the hand-written cases do not enumerate."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu

# soak runs: TFGX_FUZZ_SCALE=10 multiplies the number of seeds of every sweep
_SCALE = int(os.environ.get("TFGX_FUZZ_SCALE", "1"))


def _random_graph(rng, n, e):
    row = rng.integers(0, n, size=e).astype(np.int32)
    col = rng.integers(0, n, size=e).astype(np.int32)
    if e > 8 and rng.random() < 0.5:                       # a long row + duplicates + explicit self-loops
        k = e // 3
        row[:k] = row[0]
        col[k:k + 4] = col[k]
        row[k + 4:k + 8] = col[k + 4:k + 8]
    if rng.random() < 0.5 and n > 3:                       # some destinations without edges
        row[row == 1] = 0
    return np.stack([row, col])


@pytest.mark.parametrize("seed", range(40 * _SCALE))
def test_fuzz_segment_reduce(tfg, oracle, seed):
    from tf_geometric_amd.plan import CsrPlan, segment_reduce
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    n = int(rng.integers(1, 400))
    e = int(rng.integers(0, 4000))
    f = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 31, 32, 33, 48, 64, 65, 96, 100, 128, 130, 192, 256, 260, 300,
                        512, 520, 1100]))
    ei = _random_graph(rng, n, e)
    x = rng.standard_normal((n, f)).astype(np.float32)
    op = int(rng.integers(0, 3))
    weighted = rng.random() < 0.6
    w = rng.uniform(-1.5, 1.5, size=e).astype(np.float32) if weighted else None
    use_self = rng.random() < 0.4
    use_bias = rng.random() < 0.4
    use_add = rng.random() < 0.3
    act = int(rng.integers(0, 2))
    sc = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if use_self else None
    bias = rng.standard_normal(f).astype(np.float32) if use_bias else None
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    xd = L.as_f32(x)
    if rng.random() < 0.3 and f > 1:                       # a strided view (leading dimension > F)
        big = torch.zeros((n, f + 3), device="cuda")
        big[:, :f] = xd
        xd = big[:, :f]
    got = segment_reduce(plan, xd, op, w_csr=None if w is None else plan.edge_attr_to_csr(w),
                         self_coef=None if sc is None else L.as_f32(sc), bias=None if bias is None else L.as_f32(bias),
                         add_x=xd if use_add else None, act=act).cpu().numpy()
    # float64 restatement of the kernel's contract
    msg = x[ei[1]].astype(np.float64) * (w[:, None] if w is not None else 1.0)
    if op == 2:
        ref = np.full((n, f), -3.4028234663852886e38)
        np.maximum.at(ref, ei[0], msg)
        if sc is not None:
            ref = np.maximum(ref, sc[:, None].astype(np.float64) * x)
    else:
        ref = np.zeros((n, f))
        np.add.at(ref, ei[0], msg)
        if sc is not None:
            ref += sc[:, None].astype(np.float64) * x
        if op == 1:
            ref /= np.maximum(np.bincount(ei[0], minlength=n), 1)[:, None]
    if use_add:
        ref = x + ref
    if bias is not None:
        ref = ref + bias
    if act:
        ref = np.maximum(ref, 0)
    scale = max(1.0, float(np.abs(msg).sum(0).max()) if e else 1.0)
    assert_parity(got, ref.astype(np.float32), tol=1e-5 * scale ** 0.5 if op != 2 else 1e-5,
                  what="fuzz seg_reduce seed {} n={} e={} f={} op={}".format(seed, n, e, f, op))


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_fuzz_gat_attention(tfg, oracle, seed):
    from tf_geometric_amd.plan import CsrPlan
    from tf_geometric_amd.nn.conv.gat import gat_attention
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    n = int(rng.integers(1, 300))
    e = int(rng.integers(0, 3000))
    H = int(rng.choice([1, 2, 4, 8]))
    d = int(rng.choice([1, 2, 3, 4, 8, 16, 5]))
    dv = int(rng.choice([1, 2, 4, 8, 16, 6, 32]))
    ei = _random_graph(rng, n, e)
    Q = rng.standard_normal((n, H * d)).astype(np.float32)
    K = rng.standard_normal((n, H * d)).astype(np.float32)
    V = rng.standard_normal((n, H * dv)).astype(np.float32)
    plan = CsrPlan.build(L.as_i32(ei), n, n)
    got = gat_attention(plan, L.as_f32(Q), L.as_f32(K), L.as_f32(V), H).cpu().numpy()
    ar = np.arange(n)
    row = np.concatenate([ei[0], ar])
    col = np.concatenate([ei[1], ar])
    ref = np.zeros((n, H * dv))
    for h in range(H):
        s = (Q[row, h * d:(h + 1) * d].astype(np.float64) * K[col, h * d:(h + 1) * d]).sum(-1) / np.sqrt(d)
        m = np.full(n, -np.inf)
        np.maximum.at(m, row, s)
        p = np.exp(s - m[row])
        den = np.zeros(n)
        np.add.at(den, row, p)
        a = p / (den[row] + 1e-8)
        np.add.at(ref[:, h * dv:(h + 1) * dv], row, a[:, None] * V[col, h * dv:(h + 1) * dv])
    assert_parity(got, ref.astype(np.float32), tol=2e-5, what="fuzz gat seed {} n={} e={} H={} d={} dv={}".format(
        seed, n, e, H, d, dv))


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_fuzz_gemm(tfg, oracle, seed):
    from tf_geometric_amd.plan import gemm_bias_act
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    big = seed % 3 == 0                                    # every third case is tall enough for the row-streaming kernel
    m = int(rng.integers(32768, 50000)) if big else int(rng.integers(1, 3000))
    k = int(rng.choice([1, 3, 16, 32, 36, 60, 100, 128, 200, 256, 602])) if not big else int(rng.choice([32, 36, 60, 100, 128, 256]))
    n = int(rng.choice([1, 7, 16, 40, 64, 65, 100, 128, 200, 256, 384]))
    a = rng.standard_normal((m, k)).astype(np.float32)
    b = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32) if rng.random() < 0.5 else None
    act = int(rng.integers(0, 2))
    act_cols = int(rng.integers(0, n + 1)) if rng.random() < 0.4 else None
    got = gemm_bias_act(a, b, bias=bias, act=act, act_cols=act_cols).cpu().numpy()
    ref = a.astype(np.float64) @ b.astype(np.float64)
    if bias is not None:
        ref = ref + bias
    if act:
        c = n if act_cols is None else act_cols
        ref[:, :c] = np.maximum(ref[:, :c], 0)
    assert_parity(got, ref.astype(np.float32), tol=2e-5, what="fuzz gemm seed {} {}x{}x{}".format(seed, m, k, n))


@pytest.mark.parametrize("seed", range(32 * _SCALE))
def test_fuzz_fused_aggregate_gemm(tfg, oracle, seed):
    """tfgx_aggregate_gemm_f32 across its whole envelope, drawn at random: F in 4..128 (step 4), N in 1..256 (kernel fully or
    partly resident in LDS), sum / mean, weights, self-loop term, bias / ReLU, output into a column block, the aggregate as a
    side output, split source rows with and without the per-edge tail stream, forced hub chunking and the degree-ordered walk
    — against the float64 restatement and, bit for bit, against its own launch on the dense table."""
    from tf_geometric_amd import plan as P
    L = tfg._lib
    rng = np.random.Generator(np.random.PCG64(4000 + seed))
    n = int(rng.integers(1, 1500))
    e = int(rng.integers(0, 20000))
    f = 4 * int(rng.integers(1, 33))
    units = int(rng.choice([1, 7, 16, 40, 64, 65, 100, 128, 129, 192, 200, 256]))
    ei = _random_graph(rng, n, e)
    x = rng.standard_normal((n, f)).astype(np.float32)
    k = (rng.standard_normal((f, units)) / np.sqrt(f)).astype(np.float32)
    mean = rng.random() < 0.4
    weighted = rng.random() < 0.6
    w = rng.uniform(-1.5, 1.5, size=e).astype(np.float32) if weighted else None
    sc = rng.uniform(0.1, 1.0, size=n).astype(np.float32) if (rng.random() < 0.5 and not mean) else None
    bias = rng.standard_normal(units).astype(np.float32) if rng.random() < 0.5 else None
    act = int(rng.integers(0, 2))
    hub = rng.random() < 0.4
    if hub:
        P.HUB_THRESHOLD, P.HUB_CHUNK = int(rng.choice([8, 32, 100])), int(rng.choice([8, 16, 64]))
    try:
        plan = P.CsrPlan.build(L.as_i32(ei), n, n)
        xd, kd = L.as_f32(x), L.as_f32(k)
        w_csr = None if w is None else plan.edge_attr_to_csr(w)
        scd = None if sc is None else L.as_f32(sc)
        bd = None if bias is None else L.as_f32(bias)
        op = L.MEAN if mean else L.SUM
        assert L.require_gpu().tfgx_aggregate_gemm_fits(f, units) == 1
        side = torch.full((n, f), float("nan"), device="cuda")
        wide = torch.full((n, units + 3), 5.0, device="cuda")
        got = P.aggregate_gemm(plan, xd, op, kd, w_csr=w_csr, self_coef=scd, bias=bd, act=act, out=wide[:, 3:], agg_out=side)
        assert got is not None and bool((wide[:, :3] == 5.0).all())
        got = got.clone()
        # float64 restatement
        msg = x[ei[1]].astype(np.float64) * (w[:, None] if w is not None else 1.0)
        agg = np.zeros((n, f))
        np.add.at(agg, ei[0], msg)
        if sc is not None:
            agg += sc[:, None].astype(np.float64) * x
        if mean:
            agg /= np.maximum(np.bincount(ei[0], minlength=n), 1)[:, None]
        ref = agg @ k.astype(np.float64)
        if bias is not None:
            ref = ref + bias
        if act:
            ref = np.maximum(ref, 0)
        scale = max(1.0, float(np.abs(msg).sum(0).max()) if e else 1.0)
        tol = 2e-5 * scale ** 0.5
        what = "fuzz fused seed {} n={} e={} f={} units={} mean={} hub={}".format(seed, n, e, f, units, mean, hub)
        assert_parity(side.cpu().numpy(), agg.astype(np.float32), tol=tol, what=what + " (side output)")
        assert_parity(got.cpu().numpy(), ref.astype(np.float32), tol=tol * max(1.0, float(np.abs(k).sum(0).max())), what=what)
        # the aggregate written beside the projection = the bits of the stand-alone kernel
        assert torch.equal(side, P.segment_reduce(plan, xd, op, w_csr=w_csr, self_coef=scd))
        # split source rows (static layout), then with the per-edge tail stream: bit-identical
        if f > 32 and f % 32:
            rows = P.SplitRows.from_dense(xd)
            assert torch.equal(P.aggregate_gemm(plan, rows, op, kd, w_csr=w_csr, self_coef=scd, bias=bd, act=act), got), what
            if e:
                rows.with_edge_tail(plan)
                assert torch.equal(P.aggregate_gemm(plan, rows, op, kd, w_csr=w_csr, self_coef=scd, bias=bd, act=act), got), what
    finally:
        P.HUB_THRESHOLD, P.HUB_CHUNK = None, None
