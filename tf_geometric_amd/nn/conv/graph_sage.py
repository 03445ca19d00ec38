# coding=utf-8
"""GraphSAGE aggregators on the HIP backend — functional mirror of tf_geometric/nn/conv/graph_sage.py
(mean / sum / gcn / mean-pool / max-pool; the LSTM aggregator is an RNN, not a segment reduce: out of scope).

Reference quirks that are reproduced on purpose:
  * mean_pool / max_pool / gcn variants REPLACE a provided edge_weight by ones (:139-140, :190-191, :253-254) and
    the pool variants fail for edge_weight=None (gcn_mapper(None), :197, :260);
  * the SAME activation is applied after the pooling MLP and again at the end (:203-204 and :219-220);
  * gcn_graph_sage passes `cache` into gcn_norm_edge's `renorm` slot (:142 vs nn/conv/gcn.py:180).
Because the pool variants' edge weight is exactly 1.0, act(x[col] @ W + b) == act(x @ W + b)[col]: the reference's
per-EDGE GEMM ([E,F]x[F,4ku]) is done per NODE here, followed by the gather-reduce kernel.
"""
import torch

from ... import _lib as L
from ...activations import resolve as _resolve_act
from ...plan import (CsrPlan, segment_reduce, gemm_bias_act, l2_normalize_rows_, static_rows,
                     static_aggregate, static_aggregate_applies, gather_friendly_empty, aggregate_gemm)
from ...sparse import SparseMatrix
from .gcn import gcn_norm_adj
from ... import autograd as AG


def _combine(from_x_kernel, x, from_neigh_kernel, reduced, bias, activation, concat, normalize):
    """concat/add of x @ self_kernel and reduced @ neighbor_kernel, + bias, activation, l2-normalise
    (reference :43-58). With concat the two GEMMs write straight into the two halves of the output."""
    act, post = _resolve_act(activation)
    n = int(x.shape[0])
    ku_x, ku_n = int(from_x_kernel.shape[1]), int(from_neigh_kernel.shape[1])
    if AG.needs_grad(x, reduced, from_x_kernel, from_neigh_kernel, bias):     # training route (autograd.py)
        if concat:      # both GEMMs write into their halves of the output, bias + activation in the epilogues
            h = AG.dual_linear(x, from_x_kernel, reduced, from_neigh_kernel, bias, act)
            h = post(h) if post is not None else h
        else:
            h = AG.linear(x, from_x_kernel) + AG.linear(reduced, from_neigh_kernel)
            if bias is not None:
                h = AG.bias_add(h, bias)
            h = AG.apply_activation(h, act, post)
        if normalize:
            h = h * torch.rsqrt(torch.clamp((h * h).sum(-1, keepdim=True), min=1e-12))
        return h
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    if concat:
        h = torch.empty((n, ku_x + ku_n), dtype=torch.float32, device=x.device)
        gemm_bias_act(x, from_x_kernel, bias=None if bias_t is None else bias_t[:ku_x], act=act, out=h[:, :ku_x])
        gemm_bias_act(reduced, from_neigh_kernel, bias=None if bias_t is None else bias_t[ku_x:], act=act,
                      out=h[:, ku_x:])
    else:
        h = gemm_bias_act(x, from_x_kernel)
        h2 = gemm_bias_act(reduced, from_neigh_kernel)
        h = h + h2
        if bias_t is not None:
            h = h + bias_t
        if act == L.ACT_RELU:
            h = torch.relu_(h)
    if post is not None:
        h = post(h)
    if normalize:
        h = l2_normalize_rows_(h.contiguous())
    return h


def _neighbor_reduce(x, edge_index, edge_weight, op, cache):
    x = L.as_f32(x)
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    w_csr = AG.edge_attr_csr(plan, edge_weight, cache)                                 # :38-39
    if AG.needs_grad(x, edge_weight):
        return x, AG.aggregate(plan, x, op, w_csr)
    pre = static_aggregate(x, plan, cache, op, w_csr)      # opt-in memo: the reduce of declared-static features (layer 0)
    if pre is not None:
        return x, pre
    # declared-static input features are read in their edge-resident-tail layout (plan.static_rows)
    return x, segment_reduce(plan, static_rows(x, plan, cache), op, w_csr=w_csr)


def _concat_fused(x, edge_index, edge_weight, ws, wn, bias, activation, normalize, op, cache):
    """Inference, concat, aggregation-first (ku >= F): the neighbour half reduce(w * x[col]) @ W_neigh (+ bias, activation)
    in ONE launch straight into its half of the output (plan.aggregate_gemm: the [N, F] reduce never visits HBM), the self
    half by the GEMM.  None when the fused kernel does not take the call (shape / static layouts / hub rows)."""
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    w_csr = AG.edge_attr_csr(plan, edge_weight, cache)
    if static_aggregate_applies(x, cache):
        return None
    act, post = _resolve_act(activation)
    ku_x, ku_n = int(ws.shape[1]), int(wn.shape[1])
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    h = torch.empty((n, ku_x + ku_n), dtype=torch.float32, device=x.device)
    if aggregate_gemm(plan, static_rows(x, plan, cache), op, wn, w_csr=w_csr,
                      bias=None if bias_t is None else bias_t[ku_x:].contiguous(), act=act, out=h[:, ku_x:]) is None:
        return None
    gemm_bias_act(x, ws, bias=None if bias_t is None else bias_t[:ku_x], act=act, out=h[:, :ku_x])
    if post is not None:
        h = post(h)
    if normalize:
        h = l2_normalize_rows_(h.contiguous())
    return h


def _concat_fused_training(x, edge_index, edge_weight, ws, wn, bias, activation, normalize, op, cache):
    """The same layer body with gradients (autograd.sage_wide): the neighbour half's forward is still one launch; the
    aggregate is written beside it when d/dW_neigh is wanted.  None when the fused kernel does not take the call."""
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    if static_aggregate_applies(x, cache):
        return None
    w_csr = AG.edge_attr_csr(plan, edge_weight, cache)
    act, post = _resolve_act(activation)
    h = AG.sage_wide(plan, op, x, ws, wn, w_csr, bias, act, rows=static_rows(x, plan, cache))
    if h is None:
        return None
    if post is not None:
        h = post(h)
    if normalize:
        h = h * torch.rsqrt(torch.clamp((h * h).sum(-1, keepdim=True), min=1e-12))
    return h


def _self_neighbor_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias, activation, concat, normalize,
                        op, cache):
    """mean / sum GraphSAGE (reference :9-115).  Both reducers are linear, so
    reduce(w * x[col]) @ W_neigh == reduce(w * (x @ W_neigh)[col]): when the neighbour projection is NARROWER than the
    input (hidden layers: 256 -> units/2) the GEMM runs first and the gather moves 4*ku instead of 4*F bytes per edge —
    the same move as GCN's narrow-side aggregation; bias and activation ride in the aggregation's
    epilogue and the result lands directly in its half of the output.  Same value up to fp32 re-association."""
    x = L.as_f32(x)
    wn = L.as_f32(neighbor_kernel)
    ws = L.as_f32(self_kernel)
    F, ku_x, ku_n = int(x.shape[1]), int(ws.shape[1]), int(wn.shape[1])
    if not ku_n < F:
        if concat and not AG.needs_grad(x, edge_weight, ws, wn, bias):
            h = _concat_fused(x, edge_index, edge_weight, ws, wn, bias, activation, normalize, op, cache)
            if h is not None:
                return h
        if concat and not AG.needs_grad(edge_weight):
            h = _concat_fused_training(x, edge_index, edge_weight, ws, wn, bias, activation, normalize, op, cache)
            if h is not None:
                return h
        x, reduced = _neighbor_reduce(x, edge_index, edge_weight, op, cache)
        return _combine(ws, x, wn, reduced, bias, activation, concat, normalize)
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    w_csr = AG.edge_attr_csr(plan, edge_weight, cache)
    act, post = _resolve_act(activation)
    if AG.needs_grad(x, edge_weight, ws, wn, bias):
        if concat and not AG.needs_grad(w_csr):
            # one fused operator: both halves written in place, bias + activation in the GEMM's / the aggregation's
            # epilogue — no concat, no bias add, no activation pass
            h = AG.sage_narrow(plan, op, x, ws, wn, w_csr, bias, act)
            h = post(h) if post is not None else h
        elif concat:    # trainable edge weights: the un-fused operators carry d/dw
            bias_t = None if bias is None else L.as_f32(bias)
            a = AG.linear(x, ws, None if bias_t is None else bias_t[:ku_x], act)
            b = AG.aggregate(plan, AG.linear(x, wn, gathered=True), op, w_csr,
                             bias=None if bias_t is None else bias_t[ku_x:], act=act)
            h = torch.cat([a, b], dim=1)
            h = post(h) if post is not None else h
        else:
            h = AG.linear(x, ws) + AG.aggregate(plan, AG.linear(x, wn, gathered=True), op, w_csr)
            if bias is not None:
                h = AG.bias_add(h, bias)
            h = AG.apply_activation(h, act, post)
        if normalize:
            h = h * torch.rsqrt(torch.clamp((h * h).sum(-1, keepdim=True), min=1e-12))
        return h
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    z = gemm_bias_act(x, wn, out=gather_friendly_empty(n, ku_n, x.device))      # rows gathered next: line-friendly stride
    if concat:
        h = torch.empty((n, ku_x + ku_n), dtype=torch.float32, device=x.device)
        gemm_bias_act(x, ws, bias=None if bias_t is None else bias_t[:ku_x], act=act, out=h[:, :ku_x])
        segment_reduce(plan, z, op, w_csr=w_csr, out=h[:, ku_x:], act=act,
                       bias=None if bias_t is None else bias_t[ku_x:].contiguous())
    else:
        h = segment_reduce(plan, z, op, w_csr=w_csr, add_x=gemm_bias_act(x, ws), bias=bias_t, act=act)
    if post is not None:
        h = post(h)
    if normalize:
        h = l2_normalize_rows_(h.contiguous())
    return h


def mean_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias=None, activation=None,
                    concat=True, normalize=False, cache=None):
    """Reference: graph_sage.py:9-60."""
    return _self_neighbor_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias, activation, concat,
                               normalize, L.MEAN, cache)


def sum_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias=None, activation=None,
                   concat=True, normalize=False, cache=None):
    """Reference: graph_sage.py:64-115."""
    return _self_neighbor_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias, activation, concat,
                               normalize, L.SUM, cache)


def gcn_graph_sage(x, edge_index, edge_weight, kernel, bias=None, activation=None, normalize=False, cache=None):
    """Reference: graph_sage.py:118-161 (quirks kept, see module docstring)."""
    x = L.as_f32(x)
    n = int(x.shape[0])
    ei = L.as_i32(edge_index)
    E = int(ei.shape[1])
    if edge_weight is not None:
        edge_weight = torch.ones(E, dtype=torch.float32, device=x.device)          # :139-140
    # :142 (positional slip): `cache` lands in gcn_norm_edge's `renorm` slot, so renorm = truthiness of the dict.  This
    # package's own bookkeeping entries ("tfgx_*") must not flip it: the reference never writes to that dict here.
    renorm = bool(cache) if not isinstance(cache, dict) else any(
        not str(k[0] if isinstance(k, tuple) else k).startswith("tfgx_") for k in cache)
    # the reference caches nothing here (its `cache` lands in `renorm`); with a dict we still reuse the graph's CSR plan
    # and memoise the normalised adjacency under a PRIVATE key (same values, no radix sort / sync per forward, so the
    # layer can be captured by graph_capture.CapturedForward)
    key = ("tfgx_gcn_graph_sage_normed", edge_weight is not None, renorm)
    hit = cache.get(key) if isinstance(cache, dict) else None
    if hit is not None and hit[0] is edge_index:
        normed = hit[1]
    else:
        adj = SparseMatrix(ei, edge_weight, [n, n])
        if isinstance(cache, dict):
            adj._plan = CsrPlan.from_cache(ei, n, n, cache)
        normed = gcn_norm_adj(adj, renorm=renorm, improved=False, cache=None)
        if isinstance(cache, dict):
            cache[key] = (edge_index, normed)
    act, post = _resolve_act(activation)
    # (A_hat x) @ kernel == A_hat (x @ kernel): when the layer narrows (units < F) the GEMM runs first and the gather moves
    # units-wide rows (same value up to fp32 re-association) — F = 1433 -> 16: 9.0 -> 0.5 ms
    narrow = int(kernel.shape[1]) < int(x.shape[1])
    if AG.needs_grad(x, kernel, bias):
        if narrow:
            h = AG.aggregate(normed.plan, AG.linear(x, kernel, gathered=True), L.SUM, normed.w_csr, normed.self_coef,
                             bias=None if bias is None else L.as_f32(bias), act=act)
            h = post(h) if post is not None else h
        else:
            reduced = AG.aggregate(normed.plan, x, L.SUM, normed.w_csr, normed.self_coef)
            h = AG.apply_activation(AG.linear(reduced, kernel, bias, act), L.ACT_NONE, post)
        if normalize:
            h = h * torch.rsqrt(torch.clamp((h * h).sum(-1, keepdim=True), min=1e-12))
        return h
    if narrow:
        z = gemm_bias_act(x, kernel, out=gather_friendly_empty(n, int(kernel.shape[1]), x.device))
        h = normed.matmul(z, bias=None if bias is None else L.as_f32(bias).contiguous(), act=act)
    else:
        h = aggregate_gemm(normed.plan, x, L.SUM, kernel, w_csr=normed.w_csr, self_coef=normed.self_coef, bias=bias, act=act)
        if h is None:
            reduced = normed.matmul(x)                                              # :143-150
            h = gemm_bias_act(reduced, kernel, bias=bias, act=act)                  # :152-157
    if post is not None:
        h = post(h)
    if normalize:
        h = l2_normalize_rows_(h)
    return h


def _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                     neighbor_mlp_bias, bias, activation, concat, normalize, op, cache):
    if edge_weight is None:
        raise TypeError("edge_weight=None is not supported by the pooling aggregators "
                        "(gcn_mapper(None) fails in the reference, graph_sage.py:197/260)")
    x = L.as_f32(x)
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    act, post = _resolve_act(activation)
    if AG.needs_grad(x, neighbor_mlp_kernel, neighbor_mlp_bias, self_kernel, neighbor_kernel, bias):
        if (op == L.MAX and act == L.ACT_RELU and post is None and AG.needs_grad(neighbor_mlp_kernel, neighbor_mlp_bias)
                and AG.pool_mlp_max_applies(plan, x, neighbor_mlp_kernel)):
            # layer 0 (x is data): MLP + max as one operator whose backward goes from the destination rows straight to the
            # weight gradient — no gradient of the [N, 4 ku] hidden rows (autograd._PoolMlpMax)
            reduced = AG.pool_mlp_max(plan, x, neighbor_mlp_kernel, neighbor_mlp_bias)
        else:
            h = AG.apply_activation(AG.linear(x, neighbor_mlp_kernel, neighbor_mlp_bias, act, gathered=True), L.ACT_NONE, post)
            reduced = AG.aggregate(plan, h, op)
        return _combine(L.as_f32(self_kernel), x, L.as_f32(neighbor_kernel), reduced, bias, activation, concat,
                        normalize)
    # :199-204 per node (weight == 1).  The [N, 4 * ku] MLP rows are gathered next: written at a gather-friendly stride (units
    # 256 -> 512 columns -> rows 544 floats apart: a 2 KB row stride costs the max reduce 20 %, plan.pow2_row_stride)
    h = gemm_bias_act(x, neighbor_mlp_kernel, bias=neighbor_mlp_bias, act=act,
                      out=gather_friendly_empty(n, int(neighbor_mlp_kernel.shape[1]), x.device))
    if post is not None:
        h = post(h)
    wn, ws = L.as_f32(neighbor_kernel), L.as_f32(self_kernel)
    ku_x, ku_n = int(ws.shape[1]), int(wn.shape[1])
    if op == L.MEAN and ku_n < int(h.shape[1]):
        # the mean is linear: mean_j(h_j) @ W_neigh == mean_j(h_j @ W_neigh) (:206-208) — project the MLP rows to the
        # (4x narrower) output width first and gather THOSE: 4 * ku instead of 16 * ku bytes per edge, bias + activation in
        # the aggregation's epilogue, the result straight into its half of the output (as _self_neighbor_sage does for
        # mean / sum GraphSAGE).  max_pool cannot: max does not commute with the projection
        bias_t = None if bias is None else L.as_f32(bias).contiguous()
        z = gemm_bias_act(h, wn, out=gather_friendly_empty(n, ku_n, x.device))
        if concat:
            out = torch.empty((n, ku_x + ku_n), dtype=torch.float32, device=x.device)
            gemm_bias_act(x, ws, bias=None if bias_t is None else bias_t[:ku_x], act=act, out=out[:, :ku_x])
            segment_reduce(plan, z, L.MEAN, out=out[:, ku_x:], act=act,
                           bias=None if bias_t is None else bias_t[ku_x:].contiguous())
        else:
            out = segment_reduce(plan, z, L.MEAN, add_x=gemm_bias_act(x, ws), bias=bias_t, act=act)
        if post is not None:
            out = post(out)
        if normalize:
            out = l2_normalize_rows_(out.contiguous())
        return out
    reduced = segment_reduce(plan, h, op)                                            # :206 / :269
    return _combine(ws, x, wn, reduced, bias, activation, concat, normalize)


def mean_pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                         neighbor_mlp_bias=None, bias=None, activation=None, concat=True, normalize=False, cache=None):
    """Reference: graph_sage.py:164-225."""
    return _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                            neighbor_mlp_bias, bias, activation, concat, normalize, L.MEAN, cache)


def max_pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                        neighbor_mlp_bias=None, bias=None, activation=None, concat=True, normalize=False, cache=None):
    """Reference: graph_sage.py:228-287. An isolated node keeps float32 lowest() through the next GEMM."""
    return _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                            neighbor_mlp_bias, bias, activation, concat, normalize, L.MAX, cache)
