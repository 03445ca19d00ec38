# coding=utf-8
"""The one piece of tf_geometric.utils the hot path touches: add_self_loop_edge (GAT prologue)."""
import numpy as np
import torch

from .. import _lib as L


def add_self_loop_edge(edge_index, num_nodes, edge_weight=None, fill_weight=1.0):
    """APPEND the N diagonal edges after the input edges; existing self-loops are kept (duplicates).
    Reference: tf_geometric/utils/graph_utils.py:350-366.  (The fused GAT / GCN kernels keep this edge implicit;
    this explicit form exists for user code that calls it directly.)  numpy in -> numpy out, tensor in -> tensor out."""
    is_np = not isinstance(edge_index, torch.Tensor)
    if is_np:
        diag = np.stack([np.arange(num_nodes, dtype=np.int32)] * 2, axis=0)
        ei = np.concatenate([np.asarray(edge_index, dtype=np.int32).reshape(2, -1), diag], axis=1)
        ew = None
        if edge_weight is not None:
            ew = np.concatenate([np.asarray(edge_weight, dtype=np.float32),
                                 np.full([num_nodes], fill_weight, dtype=np.float32)])
        return ei, ew
    ar = torch.arange(num_nodes, dtype=torch.int32, device=edge_index.device)
    ei = torch.cat([edge_index.to(torch.int32), torch.stack([ar, ar])], dim=1)
    ew = None
    if edge_weight is not None:
        ew = torch.cat([L.as_f32(edge_weight, edge_index.device),
                        torch.full((num_nodes,), float(fill_weight), dtype=torch.float32, device=edge_index.device)])
    return ei, ew


# ---------------------------------------------------------------------------------------------------------------
# Edge preprocessing on the device — the step immediately before the hot path.
# Reference: tf_geometric/utils/graph_utils.py:67-125 (merge_duplicated_edge), :126-150 (convert_edge_to_upper),
# :155-212 (convert_edge_to_directed), :252-269 (remove_self_loop_edge).  numpy in -> numpy out, tensor in -> tensor.
# ---------------------------------------------------------------------------------------------------------------

def _out(t, as_numpy):
    return t.cpu().numpy() if (as_numpy and t is not None) else t


def remove_self_loop_edge(edge_index, edge_weight=None):
    """Drop edges (i, i) (reference :252-269)."""
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    mask = ei[0] != ei[1]
    w = None if edge_weight is None else L.as_f32(edge_weight)[mask]
    return _out(ei[:, mask], as_np), _out(w, as_np and not isinstance(edge_weight, torch.Tensor))


def merge_duplicated_edge(edge_index, edge_props=None, merge_modes=None):
    """Unique edges in first-occurrence order (tf.unique) + edge properties merged per unique edge with
    "sum" | "mean" | "max" | "min" (reference :67-125).  The unique pass is tfgx_merge_duplicated_edges (radix sort on
    the hash n*row+col); the merges are tfgx_segment_reduce_f32 launches."""
    import ctypes
    from ..plan import CsrPlan, segment_reduce
    if edge_props is not None and len(edge_props) > 0:
        if merge_modes is None:
            merge_modes = ["sum"] * len(edge_props)
        elif type(merge_modes) is not list:
            raise Exception("type error: merge_modes should be a list of strings")
    lib = L.require_gpu()
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    E = int(ei.shape[1])
    if E == 0:
        return _out(ei, as_np), edge_props
    n = int(ei.max().item()) + 1                                  # convert_edge_index_to_edge_hash(num_nodes=None)
    dev = ei.device
    out_row = torch.empty(E, dtype=torch.int32, device=dev)
    out_col = torch.empty(E, dtype=torch.int32, device=dev)
    uidx = torch.empty(E, dtype=torch.int32, device=dev)
    n_unique = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.tfgx_merge_edges_workspace_bytes(E, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.tfgx_merge_duplicated_edges(L.ptr(ei[0]), L.ptr(ei[1]), E, n, L.ptr(out_row), L.ptr(out_col),
                                            L.ptr(uidx), L.ptr(n_unique), L.ptr(ws), ws_bytes, L.stream_ptr()),
            "tfgx_merge_duplicated_edges")
    U = int(n_unique.item())
    unique_ei = torch.stack([out_row[:U], out_col[:U]])
    if edge_props is None:
        return _out(unique_ei, as_np), None
    plan = None
    merged = []
    for prop, mode in zip(edge_props, merge_modes):
        if prop is None:
            merged.append(None)
            continue
        if mode not in ("min", "max", "mean", "sum"):
            raise Exception("wrong merge mode: {}".format(mode))
        p = L.as_f32(prop)
        squeeze = p.dim() == 1
        p2 = p.unsqueeze(1) if squeeze else p
        if plan is None:   # message i -> segment unique_index[i]
            plan = CsrPlan.build(torch.stack([uidx, torch.arange(E, dtype=torch.int32, device=dev)]), U, E)
        if mode == "min":
            red = -segment_reduce(plan, -p2, L.MAX)
        else:
            red = segment_reduce(plan, p2, {"sum": L.SUM, "mean": L.MEAN, "max": L.MAX}[mode])
        red = red[:, 0] if squeeze else red
        merged.append(_out(red, not isinstance(prop, torch.Tensor)))
    return _out(unique_ei, as_np), merged


def convert_edge_to_upper(edge_index, edge_props=None, merge_modes=None):
    """(min(u,v), max(u,v)) per edge, duplicates merged (reference :126-150)."""
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    upper = torch.stack([torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])])
    u, props = merge_duplicated_edge(upper, edge_props, merge_modes)
    return _out(u, as_np), props


def convert_edge_to_directed(edge_index, edge_props=None, merge_modes=None):
    """[[1,3,5],[2,1,4]] -> [[1,3,5,2,1,4],[2,1,4,1,3,5]]: upper-triangular unique edges followed by their
    mirrored non-self-loop copies, properties alike (reference :155-212)."""
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    if edge_props is not None and len(edge_props) > 0 and merge_modes is None:
        merge_modes = ["sum"] * len(edge_props)
    upper, upper_props = convert_edge_to_upper(ei, edge_props, merge_modes)
    mask = upper[0] != upper[1]
    if not bool(mask.any()):
        return edge_index, edge_props                                                  # :205-207
    lower = torch.stack([upper[1][mask], upper[0][mask]])
    updated = torch.cat([upper, lower], dim=1)
    if edge_props is None:
        return _out(updated, as_np), None
    out_props = []
    for prop, up in zip(edge_props, upper_props):
        if prop is None:
            out_props.append(None)
            continue
        was_np = not isinstance(up, torch.Tensor)
        up_t = L.as_f32(up)
        out_props.append(_out(torch.cat([up_t, up_t[mask]], dim=0), was_np))
    return _out(updated, as_np), out_props


class RandomNeighborSampler(object):
    """Neighbour sampler on the device.  Mirrors tf_geometric.utils.RandomNeighborSampler
    (graph_utils.py:630-772): `sample(k=...)`, `sample(ratio=...)`, `padding=True`, and sub-graph sampling in virtual
    ids (`sampled_node_index=` one index array, or a `(rows, cols)` tuple).  The reference loops over nodes in Python
    with np.random.choice; here one kernel launch samples every row with a counter-based generator — same distribution,
    different random stream, reproducible per seed."""

    def __init__(self, edge_index, edge_weight=None):
        from ..plan import CsrPlan
        ei = L.as_i32(edge_index)
        self._numpy = not isinstance(edge_index, torch.Tensor)
        self.num_row_nodes = int(ei[0].max().item()) + 1
        self.num_col_nodes = int(ei[1].max().item()) + 1
        self.plan = CsrPlan.build(ei, self.num_row_nodes, self.num_col_nodes)
        w = torch.ones(int(ei.shape[1]), dtype=torch.float32, device=ei.device) if edge_weight is None \
            else L.as_f32(edge_weight)                                        # :635-638: weights default to ones
        self.w_csr = self.plan.edge_attr_to_csr(w)

    def _virtual_subgraph(self, sampled_node_index):
        """CSR (row_ptr, col, w) of the sub-graph in VIRTUAL ids (:689-731): virtual row i = node rows[i] (duplicates
        and any order allowed), its neighbours restricted to `cols` and renamed to their position in `cols` (for a
        repeated column id the last position wins, as the numpy assignment :698,:703 does), original neighbour order
        kept.  Index bookkeeping only (torch on the device); the sampling itself stays one kernel launch."""
        plan = self.plan
        dev = plan.col.device
        if isinstance(sampled_node_index, tuple):
            rows, cols = sampled_node_index
        else:
            rows = cols = sampled_node_index
        rows = L.as_i32(rows).long().reshape(-1)
        cols = L.as_i32(cols).long().reshape(-1)
        col_map = torch.full((self.num_col_nodes,), -1, dtype=torch.int64, device=dev)
        ok = cols < self.num_col_nodes
        # "last position wins" for a repeated id == the largest position (an indexed store would be order-dependent)
        col_map.scatter_reduce_(0, cols[ok], torch.arange(int(cols.shape[0]), device=dev)[ok], reduce="amax")
        vcol = col_map[plan.col.long()]                                         # per CSR edge
        keep = vcol >= 0
        csum = torch.zeros(plan.num_edges + 1, dtype=torch.int64, device=dev)
        csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
        kept_ptr = csum[plan.row_ptr.long()]                                    # row_ptr of the column-filtered graph
        kept_col = vcol[keep].to(torch.int32)
        kept_w = self.w_csr[keep]
        in_range = rows < self.num_row_nodes
        r = torch.where(in_range, rows, torch.zeros_like(rows))
        cnt = torch.where(in_range, kept_ptr[r + 1] - kept_ptr[r], torch.zeros_like(r))
        sub_ptr = torch.zeros(int(rows.shape[0]) + 1, dtype=torch.int64, device=dev)
        sub_ptr[1:] = torch.cumsum(cnt, 0)
        total = int(sub_ptr[-1].item())
        seg = torch.repeat_interleave(torch.arange(int(rows.shape[0]), device=dev), cnt)
        pos = kept_ptr[r][seg] + (torch.arange(total, device=dev) - sub_ptr[:-1][seg])
        return sub_ptr.to(torch.int32), kept_col[pos].contiguous(), kept_w[pos].contiguous()

    def sample(self, k=None, ratio=None, sampled_node_index=None, padding=False, seed=None):
        """`seed=None` (default, as the reference which draws from np.random on every call): a fresh 63-bit seed from
        torch's generator per call — reproducible under torch.manual_seed, different on every call.  Pass an int to pin
        one sample."""
        if k is not None and ratio is not None:
            raise Exception("k and ratio cannot be provided simultaneously")   # :674-675
        lib = L.require_gpu()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        if sampled_node_index is None:
            row_ptr, col, w_csr = self.plan.row_ptr, self.plan.col, self.w_csr
        else:
            row_ptr, col, w_csr = self._virtual_subgraph(sampled_node_index)
        n_rows = int(row_ptr.shape[0]) - 1
        deg = (row_ptr[1:] - row_ptr[:-1]).to(torch.int32)
        if k is None and ratio is None:
            cnt = deg                                                           # sample_all
        elif ratio is not None:
            cnt = torch.ceil(deg.to(torch.float64) * float(ratio)).to(torch.int32)   # :752
        elif padding:
            cnt = torch.where(deg > 0, torch.full_like(deg, int(k)), torch.zeros_like(deg))
        else:
            cnt = torch.clamp(deg, max=int(k))
        out_ptr = torch.zeros(n_rows + 1, dtype=torch.int32, device=deg.device)
        out_ptr[1:] = torch.cumsum(cnt, 0)
        total = int(out_ptr[-1].item()) if n_rows > 0 else 0
        if total == 0:
            return None, None                                                   # :767-769
        out_col = torch.empty(total, dtype=torch.int32, device=deg.device)
        out_w = torch.empty(total, dtype=torch.float32, device=deg.device)
        L.check(lib.tfgx_sample_neighbors(L.ptr(row_ptr), L.ptr(col), L.ptr(w_csr), n_rows,
                                          L.ptr(out_ptr), int(cnt.max().item()), 1 if padding else 0, int(seed),
                                          L.ptr(out_col), L.ptr(out_w), L.stream_ptr()), "tfgx_sample_neighbors")
        rows = torch.repeat_interleave(torch.arange(n_rows, dtype=torch.int32, device=deg.device), cnt.long())
        ei = torch.stack([rows, out_col])
        if not self._numpy:
            # the sample is already grouped by destination and out_ptr is its row_ptr: hand the layers a ready CSR plan
            # (CsrPlan.from_cache picks it up) instead of letting each of them sort the edge list again
            from ..plan import CsrPlan
            n_cols = self.num_col_nodes if sampled_node_index is None else int(col.max().item()) + 1 if total else 0
            ei._tfgx_plan = CsrPlan.from_sorted(out_ptr, out_col, n_cols, edge_index=ei)
        return _out(ei, self._numpy), _out(out_w, self._numpy)


# ---------------------------------------------------------------------------------------------------------------
# Laplacian / adjacency normalisation in EDGE-LIST form and sampled-subgraph re-indexing — the reference's public
# helpers around the path (ChebyNet's prologue, the sampler's epilogue).  Device tensors throughout; degrees come
# from the segment kernels (tfgx_segment_weight_sum_f32 over a CSR plan), the rest is per-edge elementwise work.
# ---------------------------------------------------------------------------------------------------------------

def _row_degree(ei, w, num_nodes):
    """deg[r] = sum of w over edges with edge_index[0] == r (tf.math.unsorted_segment_sum), on the plan kernels."""
    from ..plan import CsrPlan
    lib = L.require_gpu()
    plan = CsrPlan.build(ei, num_nodes)
    w_csr = plan.edge_attr_to_csr(w)
    deg = torch.empty(num_nodes, dtype=torch.float32, device=ei.device)
    L.check(lib.tfgx_segment_weight_sum_f32(L.ptr(plan.row_ptr), L.ptr(w_csr), num_nodes, 0.0, L.ptr(deg),
                                            L.stream_ptr()), "tfgx_segment_weight_sum_f32")
    return deg


def _finite_or_zero(t):
    return torch.where(torch.isinf(t) | torch.isnan(t), torch.zeros_like(t), t)


def get_laplacian(edge_index, num_nodes, edge_weight, normalization_type, fill_weight=1.0):
    """Edge list of the graph Laplacian as the reference writes it (utils/graph_utils.py:554-603): degrees are the row
    sums of the GIVEN edges; None: the N diagonal edges are appended with `fill_weight` and every edge (appended ones
    included) becomes deg[row] - w; 'sym': D^-1/2 A D^-1/2 then the diagonal appended with fill_weight; 'rw': D^-1 A
    then the diagonal.  numpy in -> numpy out, tensor in -> tensor out."""
    if normalization_type is not None:
        assert normalization_type in [None, 'sym', 'rw']                                   # :555-556
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    w = L.as_f32(edge_weight, ei.device)
    deg = _row_degree(ei, w, num_nodes)
    row, col = ei[0].long(), ei[1].long()
    if normalization_type is None:
        ei2, w2 = add_self_loop_edge(ei, num_nodes, w, fill_weight=fill_weight)
        out_w = _finite_or_zero(deg)[ei2[0].long()] - w2
    elif normalization_type == 'sym':
        dis = _finite_or_zero(torch.pow(deg, -0.5))
        ei2, out_w = add_self_loop_edge(ei, num_nodes, dis[row] * w * dis[col], fill_weight=fill_weight)
    else:
        dinv = _finite_or_zero(1.0 / deg)
        ei2, out_w = add_self_loop_edge(ei, num_nodes, dinv[row] * w, fill_weight=fill_weight)
    return _out(ei2, as_np), _out(out_w, as_np)


def adj_norm_edge(edge_index, num_nodes, edge_weight=None, add_self_loop=False, cache=None):
    """D^-1/2 A D^-1/2 on the edge list, D = row sums, optional appended unit self-loops first; cached under
    "adj_normed_edge" (utils/graph_utils.py:914-943)."""
    cache_key = "adj_normed_edge"
    if cache is not None:
        cached = cache.get(cache_key, None)
        if cached is not None:
            return cached
    as_np = not isinstance(edge_index, torch.Tensor)
    ei = L.as_i32(edge_index)
    w = (torch.ones(int(ei.shape[1]), dtype=torch.float32, device=ei.device) if edge_weight is None
         else L.as_f32(edge_weight, ei.device))
    if add_self_loop:
        ei, w = add_self_loop_edge(ei, num_nodes, w, fill_weight=1.0)
    dis = _finite_or_zero(torch.pow(_row_degree(ei, w, num_nodes), -0.5))
    res = _out(ei, as_np), _out(dis[ei[0].long()] * w * dis[ei[1].long()], as_np)
    if cache is not None:
        cache[cache_key] = res
    return res


class LaplacianMaxEigenvalue(object):
    """Largest-magnitude eigenvalue of the graph Laplacian (utils/graph_utils.py:884-909; ChebyNet's
    use_dynamic_lambda_max).  The reference hands a scipy matrix to ARPACK; here the operator stays on the device and an
    Arnoldi iteration runs every L @ v on the segment-reduce kernel (nn/conv/propagation.laplacian_max_eigenvalue)."""

    def __init__(self, edge_index, num_nodes, edge_weight, is_undirected=True):
        self.num_nodes = num_nodes
        self.edge_index = L.as_i32(edge_index)
        self.edge_weight = (torch.ones(int(self.edge_index.shape[1]), dtype=torch.float32, device=self.edge_index.device)
                            if edge_weight is None else L.as_f32(edge_weight, self.edge_index.device))
        self.is_undirected = is_undirected

    def __call__(self, normalization_type='sym'):
        assert normalization_type in [None, 'sym', 'rw']
        from ..nn.conv.propagation import chebynet_norm_edge, laplacian_max_eigenvalue
        if bool((self.edge_index[0] == self.edge_index[1]).any()):
            # the reference filters the WEIGHTS of self-loops but keeps the unfiltered edge_index (:896-898): with a
            # self-loop present its tf ops fail on the shape mismatch — same contract here, as an explicit error
            raise ValueError("LaplacianMaxEigenvalue: edge_index holds self-loops (the reference fails on them too: "
                             "graph_utils.py:896-898 pairs filtered weights with the unfiltered index)")
        lap = chebynet_norm_edge(self.edge_index, self.num_nodes, self.edge_weight, normalization_type,
                                 use_dynamic_lambda_max=False)
        # chebynet_norm_edge scales by 2 / lambda_max with lambda_max = 2: the plan holds the Laplacian itself
        return laplacian_max_eigenvalue(lap, normalization_type)


def reindex_sampled_edge_index(sampled_edge_index, sampled_node_index):
    """Map node ids of a sampled edge list to their positions in `sampled_node_index`; ids that were not sampled become
    -1 (the StaticHashTable default of utils/graph_utils.py:946-973).  numpy in -> numpy out, tensor in -> tensor out."""
    as_np = not isinstance(sampled_edge_index, torch.Tensor)
    ei = L.as_i32(sampled_edge_index)
    idx = L.as_i32(sampled_node_index, ei.device).long()
    hi = int(max(int(ei.max().item()) if ei.numel() else -1, int(idx.max().item()) if idx.numel() else -1)) + 1
    table = torch.full((max(hi, 1),), -1, dtype=torch.int32, device=ei.device)
    table[idx] = torch.arange(int(idx.shape[0]), dtype=torch.int32, device=ei.device)
    out = table[ei.long().clamp(min=0)]
    out = torch.where(ei < 0, torch.full_like(out, -1), out)
    return _out(out, as_np)
