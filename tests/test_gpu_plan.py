# coding=utf-8
"""The CSR-by-destination plan itself (tfgx_build_csr_by_dst; SURVEY.md K13, north_star's "edge bucketing") held to
bit-exactness against a STABLE ARGSORT of edge_index[0] — the order tf.math.unsorted_segment_* visits a segment's edges
in (ascending edge id; nn/kernel/map_reduce.py:60-70 hands `row` to the reducer unchanged).  Index work: no tolerance.

    row_ptr == [0, cumsum(bincount(row, n_dst))]      col == edge_index[1][perm]      perm == argsort(row, stable)

plus the hub chunk lists (plan.hub_info) and the transposed plan, on: a multigraph with duplicates / explicit self-loops /
empty rows, rectangular shapes, E = 0, an R-MAT graph (hub rows), and the two BASELINE shapes (arxiv vs numpy, products
vs an independent device-side stable sort)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expect(ei, n_dst):
    row = ei[0].astype(np.int64)
    perm = np.argsort(row, kind="stable").astype(np.int32)
    row_ptr = np.zeros(n_dst + 1, np.int64)
    row_ptr[1:] = np.cumsum(np.bincount(row, minlength=n_dst))
    return row_ptr.astype(np.int32), ei[1][perm].astype(np.int32), perm


def _check(plan, ei, n_dst):
    row_ptr, col, perm = _expect(ei, n_dst)
    assert plan.row_ptr.dtype == torch.int32 and plan.col.dtype == torch.int32 and plan.perm.dtype == torch.int32
    assert np.array_equal(plan.row_ptr.cpu().numpy(), row_ptr), "row_ptr"
    assert np.array_equal(plan.perm.cpu().numpy(), perm), "perm (stable: a row's edges keep the caller's order)"
    assert np.array_equal(plan.col.cpu().numpy(), col), "col"


def _hub_expect(row_ptr, thr, chunk):
    rows, ptr, begin, end, owner = [], [0], [], [], []
    for r in np.nonzero(np.diff(row_ptr.astype(np.int64)) > thr)[0]:
        b, e = int(row_ptr[r]), int(row_ptr[r + 1])
        rows.append(r)
        for s in range(b, e, chunk):
            begin.append(s)
            end.append(min(s + chunk, e))
            owner.append(r)
        ptr.append(len(begin))
    return [np.asarray(v, np.int32) for v in (rows, ptr, begin, end, owner)]


@pytest.mark.parametrize("n_dst,n_src,e,seed", [(300, 300, 5000, 1), (1, 1, 17, 2), (257, 1000, 4099, 3),
                                                 (1000, 64, 70000, 4), (65537, 65537, 200001, 5), (50, 50, 0, 6)])
def test_plan_equals_stable_argsort(tfg, n_dst, n_src, e, seed):
    from tf_geometric_amd.plan import CsrPlan
    rng = np.random.Generator(np.random.PCG64(seed))
    hi = max(1, n_dst - n_dst // 10)                                  # the last 10 % of destinations stay empty
    row = rng.integers(0, hi, size=e, dtype=np.int32)
    col = rng.integers(0, n_src, size=e, dtype=np.int32)
    if e > 100:                                                       # duplicates and (square graphs) self-loops
        row[50:100], col[50:100] = row[:50], col[:50]
        if n_dst == n_src:
            col[100:120] = row[100:120]
    ei = np.stack([row, col])
    plan = CsrPlan.build(tfg._lib.as_i32(ei), n_dst, n_src)
    _check(plan, ei, n_dst)
    if e:
        t = plan.transposed()                                        # bucketed by source: the backward / sym=False plan
        _check(t, ei[::-1], n_src)
        # edge attributes travel with perm
        w = rng.standard_normal(e).astype(np.float32)
        assert np.array_equal(plan.edge_attr_to_csr(w).cpu().numpy(), w[plan.perm.cpu().numpy()])


def test_plan_rejects_out_of_range_ids_like_tf_cpu(tfg):
    from tf_geometric_amd.plan import CsrPlan
    ei = np.array([[0, 1, 5], [1, 2, 0]], np.int32)
    with pytest.raises(Exception):
        CsrPlan.build(tfg._lib.as_i32(ei), 5, 5)
    with pytest.raises(Exception):
        CsrPlan.build(tfg._lib.as_i32(np.array([[0, 1], [1, -1]], np.int32)), 5, 5)


def test_plan_and_hub_lists_on_rmat(tfg):
    """R-MAT (0.57, 0.19, 0.19, 0.05): hub destinations take the chunked path; the chunk lists must tile every hub row's
    CSR span exactly, in order, with the plan's policy — and with a forced small threshold."""
    from tf_geometric_amd import plan as P, synthetic
    n, e = 1 << 16, 2000000
    ei_t = synthetic.rmat_edges(n, e, 7, torch.device("cuda"))
    ei = ei_t.cpu().numpy()
    plan = P.CsrPlan.build(ei_t, n, n)
    _check(plan, ei, n)
    _check(plan.transposed(), ei[::-1], n)
    row_ptr = plan.row_ptr.cpu().numpy()
    thr, chunk = P.hub_policy(plan.num_edges, n)
    hub = plan.hub_info()
    assert hub is not None and int(np.diff(row_ptr).max()) > thr
    for got, want in zip(hub, _hub_expect(row_ptr, thr, chunk)):
        assert np.array_equal(got.cpu().numpy(), want)
    lists = P.build_hub_lists(plan.row_ptr[:-1], plan.row_ptr[1:], 64, 48)
    for got, want in zip(lists, _hub_expect(row_ptr, 64, 48)):
        assert np.array_equal(got.cpu().numpy(), want)
    order = plan.row_order()
    assert order is not None
    deg = np.diff(row_ptr)
    assert np.array_equal(order.cpu().numpy(), np.argsort(-deg.astype(np.int64), kind="stable").astype(np.int32))


def test_plan_at_arxiv_shape(tfg):
    from tf_geometric_amd import synthetic
    from tf_geometric_amd.plan import CsrPlan
    n, e, _ = synthetic.WORKLOADS["arxiv"]
    ei = synthetic.synthetic_edges(n, e, seed=0)
    _check(CsrPlan.build(tfg._lib.as_i32(ei), n, n), ei, n)


def test_plan_at_products_shape_vs_independent_stable_sort(tfg):
    """123 M edges: numpy's argsort would take a minute; torch.sort(stable=True) on the device is an independent
    implementation of the same total order (key = row, ties by edge id)."""
    from tf_geometric_amd import synthetic
    from tf_geometric_amd.plan import CsrPlan
    n, e, _ = synthetic.WORKLOADS["products"]
    ei = tfg._lib.as_i32(synthetic.synthetic_edges(n, e, seed=0))
    plan = CsrPlan.build(ei, n, n)
    _, perm = torch.sort(ei[0], stable=True)
    assert torch.equal(plan.perm, perm.to(torch.int32))
    assert torch.equal(plan.col, ei[1][perm])
    counts = torch.bincount(ei[0].long(), minlength=n)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    row_ptr[1:] = torch.cumsum(counts, 0)
    assert torch.equal(plan.row_ptr, row_ptr.to(torch.int32))
