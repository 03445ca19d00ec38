# coding=utf-8
"""GCN on the HIP backend — functional mirror of tf_geometric/nn/conv/gcn.py.

act( A_hat @ (x @ kernel) + bias ):  x @ kernel on the fp32 MFMA GEMM, A_hat @ h as one fused
gather-scale-segment-sum launch over the cached CSR plan (bias and activation in its epilogue); the added
diagonal (SparseMatrix.add_diag, gcn.py:72,77,98) stays implicit as a per-row self coefficient.
"""
import torch

from ... import _lib as L
from ...activations import resolve as _resolve_act
from ...plan import segment_reduce, gemm_bias_act, static_rows, static_aggregate, gather_friendly_empty, aggregate_gemm
from ...sparse import SparseMatrix, sparse_features, sparse_dense_matmul
from ... import autograd as AG

CACHE_KEY_GCN_NORMED_ADJ_TEMPLATE = "gcn_normed_adj_{}_{}_{}_{}_{}"


def compute_cache_key(norm, add_self_loop, sym, renorm, improved):
    """Same key as the reference (gcn.py:9-20), so one cache dict serves both."""
    return CACHE_KEY_GCN_NORMED_ADJ_TEMPLATE.format(norm, add_self_loop, sym, renorm, improved)


class NormedAdj(object):
    """Normalised adjacency in plan form: w_csr[E] (CSR order) + self_coef[n] (implicit diagonal)."""

    def __init__(self, plan, w_csr, self_coef, shape):
        self.plan = plan
        self.w_csr = w_csr
        self.self_coef = self_coef
        self.shape = shape

    def matmul(self, h, num_or_size_splits=None, bias=None, act=L.ACT_NONE, cache=None):
        """`cache`: the graph's cache dict — lets repeated products with the SAME h (static input features) switch to
        the edge-resident-tail layout (plan.static_rows)."""
        return segment_reduce(self.plan, static_rows(L.as_f32(h), self.plan, cache), L.SUM, w_csr=self.w_csr,
                              self_coef=self.self_coef, bias=bias, act=act)

    def __matmul__(self, h):
        return self.matmul(h)

    def dropout(self, rate, training=False):
        """SparseMatrix.dropout on the normalised values (gcn.py:262): every stored entry — the edges AND the diagonal
        the normalisation added — is kept with probability 1 - rate and rescaled by 1 / (1 - rate) (tf.nn.dropout
        semantics).  Identity unless training; the plan is shared, only the two value arrays are new."""
        if not training or rate <= 0.0:
            return self
        if not rate < 1.0:
            raise Exception("edge_drop_rate must be in [0, 1)")
        scale = 1.0 / (1.0 - float(rate))
        w = self.w_csr * ((torch.rand_like(self.w_csr) >= rate).to(torch.float32) * scale)
        sc = self.self_coef
        if sc is not None:
            sc = sc * ((torch.rand_like(sc) >= rate).to(torch.float32) * scale)
        return NormedAdj(self.plan, w, sc, self.shape)

    def to_sparse_matrix(self):
        """COO view in the reference's layout: input edges (CSR order) followed by the N diagonal entries."""
        plan = self.plan
        rows = torch.repeat_interleave(torch.arange(plan.n_dst, dtype=torch.int32, device=plan.col.device),
                                       plan.in_degree().to(torch.int64))
        index = torch.stack([rows, plan.col])
        value = self.w_csr
        if self.self_coef is not None:
            ar = torch.arange(plan.n_dst, dtype=torch.int32, device=plan.col.device)
            index = torch.cat([index, torch.stack([ar, ar])], dim=1)
            value = torch.cat([value, self.self_coef])
        return SparseMatrix(index, value, self.shape)


def gcn_norm_adj(sparse_adj, norm="both", add_self_loop=True, sym=True, renorm=True, improved=False, cache=None):
    """Normalised adjacency for GCN (reference: gcn.py:32-130; same arguments, same cache key)."""
    lib = L.require_gpu()
    if cache is not None:
        cache_key = compute_cache_key(norm, add_self_loop, sym, renorm, improved)
        cached = cache.get(cache_key, None)
        if cached is not None:
            return cached
    if norm not in L.NORM_MODES:
        raise Exception("wrong GCN norm type: {}".format(norm))                                   # :122
    fill_weight = 2.0 if improved else 1.0                                                        # :62
    if sparse_adj.shape[0] != sparse_adj.shape[1]:                                                # :65-69
        if add_self_loop:
            raise Exception("cannot set add_self_loop=True for GCN when sparse_adj.shape[0] != sparse_adj.shape[1]")
        if sym:
            raise Exception("cannot set sym=True for GCN when sparse_adj.shape[0] != sparse_adj.shape[1]")
    if norm == "right" and sparse_adj.shape[1] > sparse_adj.shape[0]:
        # :113-119 scales column c by the ROW degree of node c: with more columns than rows the reference's tf.gather
        # runs out of range (InvalidArgumentError on TF-CPU); never read past row_deg here
        raise Exception("norm='right' indexes the row degrees by column id: sparse_adj.shape[1] must not exceed "
                        "sparse_adj.shape[0]")
    plan = sparse_adj.plan
    n = plan.n_dst
    dev = plan.row_ptr.device
    w = sparse_adj.value_csr if sparse_adj._has_value else None
    if norm == "both":
        diag = fill_weight if (add_self_loop and renorm) else 0.0                                 # :76-77
    else:
        diag = fill_weight if add_self_loop else 0.0                                              # :71-72
    row_deg = torch.empty(n, dtype=torch.float32, device=dev)
    L.check(lib.tfgx_segment_weight_sum_f32(L.ptr(plan.row_ptr), L.ptr(w), n, diag, L.ptr(row_deg), L.stream_ptr()),
            "tfgx_segment_weight_sum_f32")
    col_deg = None
    if norm == "both" and not sym:                                                                # :88-91
        tplan = plan.transposed()
        tw = tplan.edge_attr_to_csr(sparse_adj.value) if sparse_adj._has_value else None
        col_deg = torch.empty(tplan.n_dst, dtype=torch.float32, device=dev)
        L.check(lib.tfgx_segment_weight_sum_f32(L.ptr(tplan.row_ptr), L.ptr(tw), tplan.n_dst, diag, L.ptr(col_deg),
                                                L.stream_ptr()), "tfgx_segment_weight_sum_f32")
    w_out = torch.empty(plan.num_edges, dtype=torch.float32, device=dev)
    self_coef = torch.empty(n, dtype=torch.float32, device=dev)
    L.check(lib.tfgx_gcn_norm_edges_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w), n, L.ptr(row_deg),
                                        L.ptr(col_deg), L.NORM_MODES[norm], fill_weight, int(bool(add_self_loop)),
                                        int(bool(renorm)), L.ptr(w_out), L.ptr(self_coef), L.stream_ptr()),
            "tfgx_gcn_norm_edges_f32")
    normed = NormedAdj(plan, w_out, self_coef if add_self_loop else None, list(sparse_adj.shape))
    if cache is not None:
        cache[cache_key] = normed                                                                 # :125-128
    return normed


def gcn_build_cache_by_adj(sparse_adj, norm="both", add_self_loop=True, sym=True, renorm=True, improved=False,
                           override=False, cache=None):
    """Reference: gcn.py:133-153."""
    if cache is None:
        cache = {}
    elif override:
        cache[compute_cache_key(norm, add_self_loop, sym, renorm, improved)] = None
    gcn_norm_adj(sparse_adj, norm, add_self_loop, sym, renorm, improved, cache)
    return cache


def gcn_build_cache_for_graph(graph, norm="both", add_self_loop=True, sym=True, renorm=True, improved=False,
                              override=False):
    """Reference: gcn.py:156-169. `graph` needs .x/.edge_index/.edge_weight/.cache (tfg.Graph duck type)."""
    n = int(graph.x.shape[0])
    adj = SparseMatrix(graph.edge_index, getattr(graph, "edge_weight", None), [n, n])
    graph.cache = gcn_build_cache_by_adj(adj, norm=norm, add_self_loop=add_self_loop, sym=sym, renorm=renorm,
                                         improved=improved, override=override, cache=graph.cache)
    return graph.cache


def gcn_norm_edge(edge_index, num_nodes, edge_weight=None, renorm=True, improved=False, cache=None):
    """Deprecated old API of the reference (gcn.py:180-197): returns (index, value) incl. the diagonal."""
    adj = SparseMatrix(edge_index, edge_weight, [num_nodes, num_nodes])
    normed = gcn_norm_adj(adj, renorm=renorm, improved=improved, cache=cache).to_sparse_matrix()
    return normed.index, normed.value


def gcn_cache_normed_edge(graph, renorm=True, improved=False, override=False):
    """Deprecated old API (reference gcn.py:201-218): put the normalised adjacency for (renorm, improved) into
    graph.cache.  The reference's override branch calls compute_cache_key with two arguments and raises TypeError; here
    override resets the entry gcn_norm_adj would reuse (norm="both", add_self_loop=True, sym=True)."""
    if override:
        graph.cache[compute_cache_key("both", True, True, renorm, improved)] = None
    n = int(getattr(graph, "num_nodes", None) or graph.x.shape[0])
    gcn_norm_edge(graph.edge_index, n, getattr(graph, "edge_weight", None), renorm, improved, graph.cache)


def gcn_mapper(repeated_x, neighbor_x, edge_weight=None):
    from ..kernel.map_reduce import gcn_mapper as _m
    return _m(repeated_x, neighbor_x, edge_weight)


def gcn(x, sparse_adj, kernel, bias=None, activation=None, norm="both", add_self_loop=True, sym=True,
        renorm=True, improved=False, edge_drop_rate=0.0, num_or_size_splits=None, training=False, cache=None):
    """
    Functional GCN (reference: gcn.py:225-290; same arguments).

    :param x: [num_nodes, num_features]; dense, or sparse (this package's SparseMatrix / a torch sparse COO tensor)
    :param sparse_adj: SparseMatrix adjacency
    :param kernel: [num_features, num_output_features] or None (skip the GEMM, :266-267)
    :param num_or_size_splits: accepted for compatibility; it bounds memory in the reference and never changes
        the output (:274-280) — the fused kernel materialises nothing edge-sized.
    :return: [num_nodes, num_output_features]
    """
    L.require_gpu()
    xs = sparse_features(x)
    if xs is not None:
        # sparse node features (one-hot / bag-of-words rows; tf.sparse.sparse_dense_matmul, :269-270):
        # x @ W is itself a gather-scale-segment-sum with the KERNEL as the source table — the same HIP kernel
        if kernel is None:
            raise Exception("sparse node features need a kernel (the reference would propagate the SparseTensor itself)")
        x = sparse_dense_matmul(xs, kernel)
        kernel = None
    normed = gcn_norm_adj(sparse_adj, norm=norm, add_self_loop=add_self_loop, sym=sym, renorm=renorm,
                          improved=improved, cache=cache)                                         # :260
    normed = normed.dropout(edge_drop_rate, training=training)                                    # :262
    x = L.as_f32(x)
    act, post = _resolve_act(activation)
    if AG.needs_grad(x, kernel, bias):      # training: differentiable un-fused route (autograd.py)
        narrow_first = kernel is not None and int(x.shape[1]) < int(kernel.shape[1])
        h = x if (kernel is None or narrow_first) else AG.linear(x, kernel, gathered=True)
        rows = static_rows(h, normed.plan, cache) if h is x else None       # raw input features: static across epochs
        # bias + ReLU ride in the LAST kernel's epilogue (GEMM when the aggregation ran first, else the aggregation)
        bias_t = None if bias is None else L.as_f32(bias)
        if narrow_first:
            pre = static_aggregate(h, normed.plan, cache, L.SUM, normed.w_csr, normed.self_coef)   # opt-in memo (layer 0)
            fused = None
            if pre is None:
                # ONE forward launch (tfgx_aggregate_gemm_f32), on the static feature layout when x has one; the aggregate is
                # written beside it only because the kernel's gradient needs it — the projection reads it from LDS (None: shape
                # does not fit)
                fused = AG.aggregate_project(normed.plan, h, L.SUM, kernel, normed.w_csr, normed.self_coef, bias_t, act,
                                             rows=rows)
            if fused is not None:
                h = fused
            else:
                h = pre if pre is not None else AG.aggregate(normed.plan, h, L.SUM, normed.w_csr, normed.self_coef, rows=rows)
                h = AG.linear(h, kernel, bias_t, act)
        else:
            h = AG.aggregate(normed.plan, h, L.SUM, normed.w_csr, normed.self_coef, rows=rows, bias=bias_t, act=act)
        return post(h) if post is not None else h
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    if kernel is not None and int(x.shape[1]) < int(kernel.shape[1]):
        # A_hat @ (x @ W) == (A_hat @ x) @ W: gather at the NARROWER width (bytes per edge = 4*min(F, units) + 8),
        # bias + activation move into the GEMM epilogue. Same result up to fp32 re-association (inside 1e-5).
        pre = static_aggregate(x, normed.plan, cache, L.SUM, normed.w_csr, normed.self_coef)       # opt-in memo (layer 0)
        h = None
        if pre is None:
            # one launch: 64-row tiles of A_hat @ x go registers -> LDS -> MFMA against the kernel held in LDS; the
            # [N, F] aggregate never visits HBM (tfgx_aggregate_gemm_f32; None when the shape does not fit).  Source rows: x
            # itself, or its static layout (declared, or promoted on this tensor's second sighting: plan.static_rows)
            h = aggregate_gemm(normed.plan, static_rows(x, normed.plan, cache), L.SUM, kernel, w_csr=normed.w_csr,
                               self_coef=normed.self_coef, bias=bias_t, act=act)
        if h is None:
            h = gemm_bias_act(pre if pre is not None else normed.matmul(x, cache=cache), kernel, bias=bias_t, act=act)
    else:
        # the GEMM's rows are gathered next: written at a line-friendly stride (plan.gather_friendly_ld)
        h = x if kernel is None else gemm_bias_act(x, kernel, out=gather_friendly_empty(int(x.shape[0]), int(kernel.shape[1]), x.device))     # :266-272
        h = normed.matmul(h, bias=bias_t, act=act, cache=cache if kernel is None else None)       # :280-288
    return post(h) if post is not None else h
