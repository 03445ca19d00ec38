# coding=utf-8
"""CSR-by-destination plan: the per-graph state the HIP kernels run on.

The reference re-scatters on every call (tf.math.unsorted_segment_*); here the edge list is bucketed once per
graph by edge_index[0] (the aggregating node, nn/kernel/map_reduce.py:60-70) and kept in the user's ``cache``
dict exactly like the reference keeps its normalised adjacency (nn/conv/gcn.py:125-128, data/graph.py:48).
"""
import ctypes

import torch

from . import _lib as L

CACHE_KEY_PLAN = "tfgx_csr_plan"


class CsrPlan(object):
    """row_ptr[n_dst+1], col[E] (source per CSR position), perm[E] (CSR position -> original edge id)."""

    def __init__(self, row_ptr, col, perm, n_dst, n_src, num_edges):
        self.row_ptr = row_ptr
        self.col = col
        self.perm = perm
        self.n_dst = int(n_dst)
        self.n_src = int(n_src)
        self.num_edges = int(num_edges)
        self._edge_index = None   # kept only to build the transposed plan on demand (sym=False)
        self._transposed = None
        self._hub = None

    @staticmethod
    def build(edge_index, n_dst, n_src=None):
        lib = L.require_gpu()
        n_src = n_dst if n_src is None else n_src
        ei = L.as_i32(edge_index)
        if ei.dim() != 2 or ei.shape[0] != 2:
            if ei.numel() == 0:
                ei = ei.reshape(2, 0)
            else:
                raise ValueError("edge_index must have shape [2, num_edges]")
        E = int(ei.shape[1])
        dev = ei.device
        row_ptr = torch.empty(n_dst + 1, dtype=torch.int32, device=dev)
        col = torch.empty(E, dtype=torch.int32, device=dev)
        perm = torch.empty(E, dtype=torch.int32, device=dev)
        ws_bytes = lib.tfgx_csr_plan_workspace_bytes(n_dst, E)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        rc = lib.tfgx_build_csr_by_dst(L.ptr(ei[0]), L.ptr(ei[1]), E, n_dst, n_src, L.ptr(row_ptr), L.ptr(col),
                                       L.ptr(perm), L.ptr(ws), ws_bytes, L.stream_ptr())
        L.check(rc, "tfgx_build_csr_by_dst")
        plan = CsrPlan(row_ptr, col, perm, n_dst, n_src, E)
        plan._edge_index = ei
        return plan

    @staticmethod
    def from_sorted(row_ptr, col, n_src, edge_index=None):
        """Plan for edges that are ALREADY grouped by destination (row_ptr given): no sort, identity perm.  What the
        neighbour sampler produces (its out_ptr is the row_ptr)."""
        E = int(col.shape[0])
        perm = torch.arange(E, dtype=torch.int32, device=col.device)
        plan = CsrPlan(row_ptr.contiguous(), col.contiguous(), perm, int(row_ptr.shape[0]) - 1, n_src, E)
        plan._edge_index = edge_index
        plan._identity_perm = True
        return plan

    def padded_to(self, n_dst, n_src):
        """The same plan seen as an [n_dst, n_src] operator (trailing destinations without edges)."""
        if n_dst == self.n_dst and n_src == self.n_src:
            return self
        if n_dst < self.n_dst or n_src < self.n_src:
            return None
        tail = self.row_ptr[-1:].expand(n_dst - self.n_dst)
        plan = CsrPlan(torch.cat([self.row_ptr, tail]).contiguous(), self.col, self.perm, n_dst, n_src, self.num_edges)
        plan._edge_index = self._edge_index
        plan._identity_perm = getattr(self, "_identity_perm", False)
        return plan

    @staticmethod
    def from_cache(edge_index, n_dst, n_src=None, cache=None, key=CACHE_KEY_PLAN):
        if cache is not None:
            plan = cache.get(key, None)
            if plan is not None:
                return plan
        attached = getattr(edge_index, "_tfgx_plan", None)      # a producer that already knows the CSR (the sampler)
        plan = attached.padded_to(n_dst, n_dst if n_src is None else n_src) if attached is not None else None
        if plan is None:
            plan = CsrPlan.build(edge_index, n_dst, n_src)
        if cache is not None:
            cache[key] = plan
        return plan

    def edge_attr_to_csr(self, attr):
        """attr[E] or attr[E, k] in the caller's edge order -> CSR order (None stays None)."""
        if attr is None:
            return None
        lib = L.require_gpu()
        a = L.as_f32(attr).contiguous()
        width = 1 if a.dim() == 1 else int(a.shape[1])
        if a.shape[0] != self.num_edges:
            raise ValueError("edge attribute has {} rows, graph has {} edges".format(a.shape[0], self.num_edges))
        if getattr(self, "_identity_perm", False):
            return a
        out = torch.empty_like(a)
        L.check(lib.tfgx_permute_rows_f32(L.ptr(a), L.ptr(self.perm), self.num_edges, width, L.ptr(out),
                                          L.stream_ptr()), "tfgx_permute_rows_f32")
        return out

    def transposed(self):
        """Plan bucketed by edge_index[1] (column sums for sym=False, nn/conv/gcn.py:88)."""
        if self._transposed is None:
            ei = self._edge_index
            self._transposed = CsrPlan.build(torch.stack([ei[1], ei[0]]), self.n_src, self.n_dst)
        return self._transposed

    def in_degree(self):
        return (self.row_ptr[1:] - self.row_ptr[:-1])

    def row_order(self):
        """Rows sorted by descending length (int32 [n_dst]) when the plan is skewed — longest row more than 8x the mean —
        else None.  Kernels that give every row a lane group walk the rows in this order so that the rows sharing a wave
        have similar lengths (GAT layer on R-MAT graphs: 13-16 %); results do not depend on it.  Computed once per plan."""
        if getattr(self, "_row_order", None) is None:
            if torch.cuda.is_current_stream_capturing():
                return None      # the skew test synchronises (deg.max().item()): never inside a hipGraph capture —
                                 # results do not depend on the order, and the next eager call computes and keeps it
            order = False
            if self.n_dst > 0 and self.num_edges > 0:
                deg = self.in_degree()
                if int(deg.max().item()) > 8 * max(self.num_edges / float(self.n_dst), 1.0):
                    order = torch.argsort(deg, descending=True, stable=True).to(torch.int32).contiguous()
            self._row_order = order
        return None if self._row_order is False else self._row_order

    def hub_info(self):
        """Chunk lists for destinations with more than hub_threshold in-edges (power-law "hubs"), or None.
        Small control-plane metadata, computed once per plan:
        (hub_rows, chunk_ptr, chunk_begin, chunk_end, chunk_row)."""
        if self._hub is None:
            if torch.cuda.is_current_stream_capturing():
                return None      # building the lists synchronises; long rows then run inline (same sums, slower)
            thr, chunk = hub_policy(self.num_edges, self.n_dst)
            self.hub_threshold = thr
            self._hub = build_hub_lists(self.row_ptr[:-1], self.row_ptr[1:], thr, chunk) or False
        return self._hub or None

    def source_blocks(self, num_blocks):
        """The plan with every destination row's edges stably partitioned by SOURCE BLOCK (block b = sources
        [b * ceil(n_src / KB), (b + 1) * ...)): (rpk int32 [n_dst * KB + 1], col_k int32 [E]) — row r's edges
        from block b sit at positions [rpk[r * KB + b], rpk[r * KB + b + 1]) of col_k (a row's edges keep their order inside a block).  KB chained
        launches over the blocks (row_begin = rpk + b, row_end = rpk + b + 1, rp_stride = KB) each gather rows of one block
        only (nn/conv/gat.py: dense graphs whose K | V table fits the L2 block by block).  Built once per (plan, KB) with
        device sorts; E * 12 bytes while building, E * 4 + n_dst * KB * 4 kept."""
        KB = int(num_blocks)
        memo = self.__dict__.setdefault("_source_blocks", {})
        if KB not in memo:
            if torch.cuda.is_current_stream_capturing():
                return None
            dev = self.col.device
            blk = max(-(-self.n_src // KB), 1)
            rows = torch.repeat_interleave(torch.arange(self.n_dst, device=dev, dtype=torch.int64), self.in_degree().long())
            key = rows * KB + torch.div(self.col.long(), blk, rounding_mode="floor")
            del rows
            order = torch.argsort(key, stable=True)
            rpk = torch.zeros(self.n_dst * KB + 1, dtype=torch.int32, device=dev)
            rpk[1:] = torch.cumsum(torch.bincount(key, minlength=self.n_dst * KB), 0).to(torch.int32)
            del key
            memo[KB] = (rpk, self.col[order].contiguous())
        return memo[KB]

    def hub_order_slot(self):
        """int32 [n_hub]: the slot in hub_rows of walk-order entry i (row_order sorts by descending length, so its first
        n_hub entries ARE the hub rows) — tfgx_reduce_args.hub_order_slot; None without hub rows / walk order."""
        if getattr(self, "_hub_order_slot", None) is None:
            hub, order = self.hub_info(), self.row_order()
            slot = False
            if hub is not None and order is not None:
                n_hub = int(hub[0].shape[0])
                slot = torch.searchsorted(hub[0], order[:n_hub].contiguous()).to(torch.int32).contiguous()
            self._hub_order_slot = slot
        return None if self._hub_order_slot is False else self._hub_order_slot


def build_hub_lists(span_begin, span_end, threshold, chunk):
    """Cut every span [begin[r], end[r]) longer than `threshold` into chunks of `chunk` CSR positions.
    -> (rows, chunk_ptr, chunk_begin, chunk_end, chunk_row) int32 device tensors, or None if no span is that long."""
    length = (span_end - span_begin)
    hub_rows = torch.nonzero(length > threshold).flatten()
    if hub_rows.numel() == 0:
        return None
    start = span_begin[hub_rows].to(torch.int64)
    d = length[hub_rows].to(torch.int64)
    n_chunks = (d + chunk - 1) // chunk
    chunk_ptr = torch.zeros(hub_rows.numel() + 1, dtype=torch.int64, device=length.device)
    chunk_ptr[1:] = torch.cumsum(n_chunks, 0)
    total = int(chunk_ptr[-1].item())
    owner = torch.repeat_interleave(torch.arange(hub_rows.numel(), device=length.device), n_chunks)
    k = torch.arange(total, device=length.device) - chunk_ptr[owner]
    begin = start[owner] + k * chunk
    end = torch.minimum(begin + chunk, start[owner] + d[owner])
    hub_rows = hub_rows.to(torch.int32)
    return (hub_rows.contiguous(), chunk_ptr.to(torch.int32).contiguous(), begin.to(torch.int32).contiguous(),
            end.to(torch.int32).contiguous(), hub_rows[owner].contiguous())


HUB_THRESHOLD = None   # override (tools/bench_sweep.py): rows with more in-edges than this take the chunked path
HUB_CHUNK = None       # override: edges per chunk


def hub_policy(num_edges, n_dst):
    """(threshold, chunk) for the chunked hub path of tfgx_segment_reduce_f32.

    A row is walked sequentially by ONE lane group, so on skewed graphs long rows become a latency-bound tail
    (R-MAT 2^21 nodes / 61 M edges, measured on MI355X: threshold 2048 -> 9.9 ms, 512 -> 5.2 ms, 128 -> 4.8 ms).
    Rows above ~4x the average in-degree are therefore cut into chunks; on near-regular graphs (uniform products /
    Reddit shapes) no row exceeds that and the path is never taken."""
    if HUB_THRESHOLD is not None:
        return int(HUB_THRESHOLD), int(HUB_CHUNK or max(HUB_THRESHOLD // 2, 64))
    avg = float(num_edges) / max(int(n_dst), 1)
    thr = 128
    while thr < 4.0 * avg and thr < 2048:
        thr *= 2
    return thr, max(128, thr // 2)


class SplitRows(object):
    """Source rows stored as main[n, f_main] (f_main a multiple of 32 floats: whole 128-byte lines) + tail[n, F-f_main].
    A 400-byte row (F = 100) at a 400-byte stride always straddles FOUR 128-byte lines (512 bytes fetched); split,
    the gather fetches three full lines from HBM and 16 bytes from a 38 MB array that stays cache-resident."""

    def __init__(self, main, tail):
        self.main, self.tail = main, tail
        self.shape = (int(main.shape[0]), int(main.shape[1]) + int(tail.shape[1]))
        self.device = main.device
        self.edge_tail, self.edge_plan = None, None

    def with_edge_tail(self, plan):
        """Also keep the tail columns PER EDGE of `plan` (edge_tail[i] = tail[col[i]], 4*(F - f_main) bytes per edge,
        16 B at F = 100): launches on that plan then stream the tail next to col / w instead of gathering it, so a
        400-byte source row costs three line requests instead of four.  Worth its memory when the same features are
        aggregated many times over the same graph — the dataset's input features (layer 0, every epoch); it has to be
        rebuilt whenever the features change."""
        self.edge_tail = gather_rows(self.tail, plan.col)
        self.edge_plan = plan
        return self

    @staticmethod
    def wanted(n, F):
        """Split only when it removes over-fetch: F not a multiple of 32, a 4..28-float tail, and a feature matrix
        far larger than the caches (otherwise the single array is already cache-resident)."""
        return F % 4 == 0 and F > 32 and F % 32 != 0 and F <= 128 and n * F * 4 > (512 << 20)

    @staticmethod
    def from_dense(x, out=None):
        lib = L.require_gpu()
        x, ldx = L.row_major_2d(x)
        n, F = int(x.shape[0]), int(x.shape[1])
        f_main = (F // 32) * 32
        if out is None:
            out = SplitRows(torch.empty((n, f_main), dtype=torch.float32, device=x.device),
                            torch.empty((n, F - f_main), dtype=torch.float32, device=x.device))
        L.check(lib.tfgx_split_rows_f32(L.ptr(x), ldx, n, F, f_main, L.ptr(out.main), f_main, L.ptr(out.tail),
                                        F - f_main, L.stream_ptr()), "tfgx_split_rows_f32")
        return out


USE_ROW_ORDER = True      # degree-ordered row walk of the segment-reduce kernel on skewed plans (developer A/B switch)
ROW_ORDER_MAX_F = 128     # ... applied up to this width (same-box A/B on R-MAT graphs, profiles/r04_ab_row_order.jsonl: see segment_reduce)

CACHE_KEY_STATIC = "tfgx_static_features"      # user opt-in: cache[CACHE_KEY_STATIC] = the static feature tensor
CACHE_KEY_STATIC_ROWS = "tfgx_static_rows"      # the layout built for it: (key, SplitRows | None, x, plan, bytes)


STATIC_STATS = {"hits": 0, "builds": 0}         # diagnostics: how often the prepared layout was used / (re)built


def _static_key(x, plan):
    return (x.data_ptr(), x._version, int(x.shape[0]), int(x.shape[1]), id(plan))


def _build_static_rows(x, plan, cache):
    n, F = int(x.shape[0]), int(x.shape[1])
    rows, nbytes = None, 0
    if SplitRows.wanted(n, F):
        rows = SplitRows.from_dense(x).with_edge_tail(plan)
        nbytes = 4 * (rows.main.numel() + rows.tail.numel() + rows.edge_tail.numel())
    cache[CACHE_KEY_STATIC_ROWS] = (_static_key(x, plan), rows, x, plan, nbytes)    # holds x and plan alive
    STATIC_STATS["builds"] += 1
    return rows, nbytes


CACHE_KEY_STATIC_AGG = "tfgx_static_aggregated"   # opt-in memo of layer-0 aggregations of the static features


def prepare_static_features(x, edge_index, cache, num_nodes=None, cache_aggregation=False):
    """EXPLICIT opt-in to the static-feature layout: declares that the tensor `x` is the graph's
    static input features — aggregated again and again over the same graph (layer 0 of a model, every epoch) and not
    written to in between — and builds, now, the SplitRows + edge-resident-tail form the aggregation kernel then uses
    for `x` (bit-identical results, one 128-byte line request fewer per edge at widths like F = 100).

    Nothing is ever built without this call (or `cache["tfgx_static_features"] = x`, which builds lazily on the first
    eager aggregation of x): the layout is derived from feature VALUES and costs 4*N*F + 4*E*(F mod 32) bytes, so it
    is the caller's decision.  Because it is built here, eagerly, a later hipGraph capture (CapturedForward) of a model
    that closes over the same `x` replays the fast layout.  Contract: while opted in, change x only through torch (a
    new tensor, or an in-place op — the version counter invalidates the layout and it is rebuilt); writes that bypass
    the counter (`x.data`, foreign pointers) require `release_static_features(cache)` first.

    cache_aggregation=True additionally lets layer 0 keep its AGGREGATED input — A_hat @ x (GCN) or the neighbour
    mean / sum of x (GraphSAGE) is the same tensor in every step while neither x nor the edge weights change, so it is
    computed once and reused (`static_aggregate`; +4*N*F bytes per distinct aggregation).  Training-time input or edge
    dropout changes the operands every step, so nothing is reused then.

    :return: dict(bytes=..., layout="edge_tail" | "dense", f_main=..., f_tail=...) — what was built and what it costs."""
    if cache is None:
        raise ValueError("prepare_static_features needs the graph's cache dict")
    x = L.as_f32(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("static features must be a contiguous [num_nodes, num_features] float32 tensor")
    n = int(x.shape[0]) if num_nodes is None else int(num_nodes)
    plan = edge_index if isinstance(edge_index, CsrPlan) else CsrPlan.from_cache(edge_index, n, int(x.shape[0]), cache)
    cache[CACHE_KEY_STATIC] = x
    cache.pop(CACHE_KEY_STATIC_AUTO, None)          # an explicit declaration: rebuilt on a version bump, never dropped
    if cache_aggregation:
        cache[CACHE_KEY_STATIC_AGG] = {}
    else:
        cache.pop(CACHE_KEY_STATIC_AGG, None)
    rows, nbytes = _build_static_rows(x, plan, cache)
    F = int(x.shape[1])
    return dict(bytes=nbytes, layout="edge_tail" if rows is not None else "dense", tensor=x,
                f_main=(F // 32) * 32 if rows is not None else F, f_tail=F % 32 if rows is not None else 0)


def release_static_features(cache):
    """Undo prepare_static_features (or an automatic promotion): frees the layout; later aggregations read x itself."""
    cache.pop(CACHE_KEY_STATIC_AUTO, None)
    cache.pop(CACHE_KEY_STATIC, None)
    cache.pop(CACHE_KEY_STATIC_ROWS, None)
    cache.pop(CACHE_KEY_STATIC_AGG, None)


def _declared_static(x, cache):
    if cache is None or not isinstance(x, torch.Tensor):
        return False
    opt = cache.get(CACHE_KEY_STATIC, None)
    if opt is None or opt is False or not isinstance(opt, torch.Tensor):
        return False
    return (opt.data_ptr() == x.data_ptr() and opt.shape == x.shape and opt.dtype == x.dtype
            and opt.stride() == x.stride())


def static_aggregate(x, plan, cache, op, w_csr=None, self_coef=None):
    """segment_reduce(plan, x, op, w_csr, self_coef) for the graph's declared-static features — computed ONCE per
    (x contents, plan, op, weights) when the caller opted in with prepare_static_features(..., cache_aggregation=True),
    returned from the memo afterwards; None when the memo does not apply (the caller then aggregates as usual).
    The result must be treated as read-only."""
    store = cache.get(CACHE_KEY_STATIC_AGG) if cache is not None else None
    if store is None or not _declared_static(x, cache):
        return None
    if any(t is not None and t.requires_grad for t in (x, w_csr, self_coef)):
        return None
    if torch.cuda.is_current_stream_capturing():
        return None      # a captured graph must contain the aggregation itself: a replay cannot see x's version counter
    ident = lambda t: None if t is None else (t.data_ptr(), t._version, int(t.numel()))      # noqa: E731
    key = (_static_key(x, plan), ident(w_csr), ident(self_coef))
    hit = store.get(op)
    if hit is not None and hit[0] == key:
        STATIC_STATS["agg_hits"] = STATIC_STATS.get("agg_hits", 0) + 1
        return hit[1]
    out = segment_reduce(plan, static_rows(x, plan, cache), op, w_csr=w_csr, self_coef=self_coef)
    store[op] = (key, out, x, w_csr, self_coef)            # the operands stay alive: their addresses cannot be recycled
    return out


def static_aggregate_applies(x, cache):
    """Would static_aggregate serve / compute the memo for x?  (No side effects: callers that have a cheaper route for
    NON-memoised features ask first.)"""
    store = cache.get(CACHE_KEY_STATIC_AGG) if cache is not None else None
    return store is not None and _declared_static(x, cache) and not torch.cuda.is_current_stream_capturing()


# AUTOMATIC promotion to the static layout (round 4): a drop-in user never calls prepare_static_features — the
# reference's own epoch loop (demo/demo_gcn.py:68-77) passes the SAME feature tensor to layer 0 in every step.  When a layer
# meets the same tensor again (same live storage, same torch version counter, same plan) after it has already been
# aggregated once, and the layout would remove over-fetch (SplitRows.wanted) and fits the budget, the layout is built then
# and used from that call on — under exactly the contract prepare_static_features states: a torch-visible write (version
# counter) drops it again (a tensor that changes every step is never promoted); writes that bypass the counter (x.data,
# foreign pointers) are not seen.  TFGX_STATIC_LAYOUT=explicit (or plan.AUTO_STATIC_LAYOUT = False) turns promotion off;
# TFGX_STATIC_LAYOUT_BUDGET = bytes the layout may take (default: 10 % of the HBM free at that moment).
import os as _os
import weakref as _weakref

AUTO_STATIC_LAYOUT = _os.environ.get("TFGX_STATIC_LAYOUT", "auto") != "explicit"
CACHE_KEY_STATIC_SEEN = "tfgx_static_seen"        # {data_ptr: (weakref(x), key, launch counter at first sight)}
CACHE_KEY_STATIC_AUTO = "tfgx_static_auto"        # True while the opt-in in this cache was made by the promotion
CACHE_KEY_STATIC_DISTRUST = "tfgx_static_distrust"  # weakref(storage) whose promoted copy went stale behind the version counter
_AGG_LAUNCHES = [0]                                # aggregation launches so far (segment_reduce / aggregate_gemm)


_AUTO_SUPPRESSED = [0]                             # > 0 inside no_auto_promotion()


class no_auto_promotion(object):
    """Context: no tensor is promoted while it is active (CapturedForward's warm-up: its static INPUT buffers are rewritten
    before every replay — the opposite of static features)."""

    def __enter__(self):
        _AUTO_SUPPRESSED[0] += 1
        return self

    def __exit__(self, *exc):
        _AUTO_SUPPRESSED[0] -= 1
        return False


def written_by_kernel(t):
    """Tell torch that a launch of this library wrote into `t` through its raw pointer (a caller-provided `out=`): bumps the
    tensor's version counter, which every contents-keyed memo here (static layout, CSR-ordered edge weights, transposed
    weights) relies on."""
    if isinstance(t, torch.Tensor):
        torch.autograd.graph.increment_version(t)
    return t


def static_layout_budget_bytes():
    env = _os.environ.get("TFGX_STATIC_LAYOUT_BUDGET")
    if env is not None:
        return int(float(env))
    free, _ = torch.cuda.mem_get_info()
    return free // 10


def _auto_promote(x, plan, cache):
    """Second sighting of the same tensor (see above) -> declare it static in `cache`; True when promoted."""
    if (not AUTO_STATIC_LAYOUT or _AUTO_SUPPRESSED[0] > 0 or x.dim() != 2 or not x.is_contiguous() or x.requires_grad or x.dtype != torch.float32
            or not x.is_cuda or not SplitRows.wanted(int(x.shape[0]), int(x.shape[1]))
            or torch.cuda.is_current_stream_capturing()):
        return False
    bad = cache.get(CACHE_KEY_STATIC_DISTRUST)
    if bad is not None and bad() is x.untyped_storage():
        return False                                   # caught once with a write the version counter missed (static_rows)
    seen = cache.setdefault(CACHE_KEY_STATIC_SEEN, {})
    key = _static_key(x, plan)
    # "the same tensor" = the same live STORAGE (layers see a fresh .detach() view of the caller's tensor on every call;
    # torch keeps one Python object per storage while it is alive, so a weak reference tells a tensor that was freed and
    # whose address the allocator handed out again — a hidden activation of the next step — from the caller's matrix)
    store = x.untyped_storage()
    hit = seen.get(x.data_ptr())
    if hit is None or hit[0]() is not store or hit[1] != key:
        if len(seen) >= 8:
            seen.clear()
        seen[x.data_ptr()] = (_weakref.ref(store), key, _AGG_LAUNCHES[0])
        return False
    if _AGG_LAUNCHES[0] == hit[2]:
        return False                                   # still the first layer call (it looks the tensor up more than once)
    F = int(x.shape[1])
    need = 4 * (int(x.shape[0]) * F + plan.num_edges * (F % 32))
    if need > static_layout_budget_bytes():
        return False
    seen.pop(x.data_ptr(), None)
    cache[CACHE_KEY_STATIC] = x
    cache[CACHE_KEY_STATIC_AUTO] = True
    STATIC_STATS["auto_promotions"] = STATIC_STATS.get("auto_promotions", 0) + 1
    STATIC_STATS["auto_promoted_bytes"] = need
    import logging
    logging.getLogger("tf_geometric_amd").info(
        "static feature layout: tensor [%d, %d] seen unchanged twice -> split + edge-tail copy of %.2f GB held in the graph's "
        "cache (TFGX_STATIC_LAYOUT=explicit turns this off; tfg.release_static_features(cache) frees it)",
        int(x.shape[0]), F, need / 1e9)
    return True


# FAIL-SAFE of the automatic promotion (round 5): a layout this library built on its own initiative is a COPY of feature values
# the caller never promised to leave alone, and torch's version counter does not see every write (x.data.copy_, DLPack, a
# foreign kernel, the tf.load_op_library route).  Before such a layout is served, a sample of its rows is compared bit for bit
# with x on the device (tfgx_split_rows_verify_f32: rows 0 and n - 1 and TFGX_STATIC_VERIFY_ROWS - 2 rows drawn afresh for every
# call — 4096 by default, "all" compares every row); a mismatch demotes the layout and the call reads x itself.  A bulk
# rewrite of the table is caught on the first call after it, a write to a few rows as soon as a draw covers one of them
# (every row with TFGX_STATIC_VERIFY_ROWS=all, at the cost of one more pass over x); layouts the caller DECLARED
# (prepare_static_features) are the caller's contract and are not re-checked.  Costs one small launch and one 4-byte read-back
# per served call (~20 us beside a 7 ms aggregation at products shape: automatic_promotion.ms_per_call in the bench line).
_VERIFY_CALLS = [0]


def _verify_rows_setting():
    v = _os.environ.get("TFGX_STATIC_VERIFY_ROWS", "4096")
    return -1 if v == "all" else max(int(v), 0)


def _promoted_layout_is_current(x, rows):
    """True when the sampled rows of the promoted layout `rows` still equal x (always True with the check switched off)."""
    samples = _verify_rows_setting()
    if samples == 0:
        return True
    n, F = int(x.shape[0]), int(x.shape[1])
    if samples < 0:
        samples = n
    lib = L.require_gpu()
    flag = torch.empty(1, dtype=torch.int32, device=x.device)
    _VERIFY_CALLS[0] += 1
    L.check(lib.tfgx_split_rows_verify_f32(L.ptr(x), int(x.stride(0)), n, F, int(rows.main.shape[1]), L.ptr(rows.main), int(rows.main.stride(0)),
                                           L.ptr(rows.tail), int(rows.tail.stride(0)), samples,
                                           (0x5DEECE66D * _VERIFY_CALLS[0] + 11) & 0xFFFFFFFFFFFFFFFF, L.ptr(flag), L.stream_ptr()),
            "tfgx_split_rows_verify_f32")
    STATIC_STATS["verifications"] = STATIC_STATS.get("verifications", 0) + 1
    return int(flag.item()) == 0


def static_rows(x, plan, cache):
    """`x`, or its SplitRows + edge-resident-tail form when this tensor is the graph's static feature matrix: declared by
    the caller (`prepare_static_features`, or `cache["tfgx_static_features"] = x`) or promoted automatically on its second
    aggregation (see AUTO_STATIC_LAYOUT above).  A torch-visible change of x (version counter) rebuilds a DECLARED layout
    outside hipGraph capture and falls back to x inside it (building allocates); an automatically promoted one is dropped."""
    if cache is None or not isinstance(x, torch.Tensor):
        return x
    opt = cache.get(CACHE_KEY_STATIC, None)
    auto = bool(cache.get(CACHE_KEY_STATIC_AUTO))
    if auto and torch.cuda.is_current_stream_capturing():
        # a hipGraph replay cannot see x's version counter: only a layout the caller DECLARED (prepare_static_features: "x is
        # not written to") may be baked into a captured sequence — a promoted one is this library's guess, and the static
        # buffers of a captured model are exactly the tensors callers overwrite between replays
        return x
    # "the same tensor": same storage window (views made by .detach() / as_f32 share it); opt is kept alive by the
    # cache entry, so the address cannot have been recycled
    same = isinstance(opt, torch.Tensor) and (opt.data_ptr() == x.data_ptr() and opt.shape == x.shape and
                                              opt.dtype == x.dtype and opt.stride() == x.stride())
    if not same:
        # no static tensor yet — or an AUTOMATICALLY promoted one that is not this tensor: a candidate seen twice takes its
        # place (the caller moved on to another feature matrix; the old layout must not be kept alive for ever).  A tensor
        # the caller DECLARED static is never displaced.
        if (opt is None or (auto and isinstance(opt, torch.Tensor))) and _auto_promote(x, plan, cache):
            cache.pop(CACHE_KEY_STATIC_ROWS, None)
            cache.pop(CACHE_KEY_STATIC_AGG, None)
        else:
            return x
    hit = cache.get(CACHE_KEY_STATIC_ROWS)
    if hit is not None and hit[0] == _static_key(x, plan):
        if hit[1] is not None and cache.get(CACHE_KEY_STATIC_AUTO) and not _promoted_layout_is_current(x, hit[1]):
            # written to behind the version counter: the promoted copy is stale — drop it, read x, start counting again
            release_static_features(cache)
            STATIC_STATS["auto_demotions"] = STATIC_STATS.get("auto_demotions", 0) + 1
            STATIC_STATS["stale_copies_caught"] = STATIC_STATS.get("stale_copies_caught", 0) + 1
            # its owner writes behind the version counter: "seen unchanged twice" cannot be established for this storage any
            # more — it is never promoted again in this cache (an explicit prepare_static_features still works)
            cache[CACHE_KEY_STATIC_DISTRUST] = _weakref.ref(x.untyped_storage())
            return x
        if hit[1] is not None:
            STATIC_STATS["hits"] += 1
        return x if hit[1] is None else hit[1]
    if torch.cuda.is_current_stream_capturing() or x.dim() != 2 or not x.is_contiguous():
        return x
    if hit is not None and cache.get(CACHE_KEY_STATIC_AUTO):
        # promoted automatically and now written to: not static after all — drop the layout, start counting again
        release_static_features(cache)
        STATIC_STATS["auto_demotions"] = STATIC_STATS.get("auto_demotions", 0) + 1
        _auto_promote(x, plan, cache)
        return x
    rows, _ = _build_static_rows(x, plan, cache)
    return x if rows is None else rows


def edge_weight_csr(plan, edge_weight, cache=None):
    """edge_weight (caller's edge order) -> CSR order, memoised in the graph's `cache` dict for as long as the SAME
    array object is passed (the reference caches its normalised adjacency under the same contract: one cache per
    graph, inputs not mutated in place — nn/conv/gcn.py:125-128)."""
    if edge_weight is None:
        return None
    ver = edge_weight._version if isinstance(edge_weight, torch.Tensor) else None     # a torch in-place write is seen
    if cache is not None:
        hit = cache.get("tfgx_edge_weight_csr")
        if hit is not None and hit[0] is edge_weight and hit[2] is plan and hit[3] == ver:
            return hit[1]
    w_csr = plan.edge_attr_to_csr(edge_weight)
    if cache is not None:
        cache["tfgx_edge_weight_csr"] = (edge_weight, w_csr, plan, ver)
    return w_csr


RELAY_POW2_TABLES = True     # developer A/B switch


def relaid_for_gather(x, ldx, plan_is_skewed, dry_run=False):
    """A caller's dense [n, F] table whose row stride is a power of two (pow2_row_stride) -> the same rows 32 floats further
    apart, when one strided copy (read + write of the table) costs less than it saves: strides of 2 KB and more on any graph
    (uniform products graph, F = 512: 43.8 -> 36.7 ms for a 2 ms copy), 512 bytes and more on a power-law plan (R-MAT, F = 256:
    25.6 -> 19.3 ms for 1 ms).  Only tables beyond the caches (> 512 MB) and only when the copy fits half of the free memory.
    Returns (x, ldx) unchanged otherwise.

    Round 6: the copy is MEMOISED per table (storage window + torch version counter; the two most recent tables) — a model
    that aggregates the same hidden table in several launches, or the same input every step, pays the 1-2 ms copy and its
    allocation once.  It is a copy of values the caller never promised to leave alone, so it is served under the fail-safe of
    the promoted static layouts: before every use sampled rows (TFGX_STATIC_VERIFY_ROWS) are compared bit for bit with x on
    the device; a write the version counter missed drops the copy and the call re-copies.  Inside a hipGraph capture nothing
    is memoised or served from the memo (a replay could not see the table change): the copy is one more captured launch."""
    F = int(x.shape[1])
    if not (RELAY_POW2_TABLES and ldx == F and pow2_row_stride(F) and (F >= 512 or plan_is_skewed)):
        return x, ldx
    n = int(x.shape[0])
    need = 4 * n * (F + 32)
    if 4 * n * F <= (512 << 20):
        return x, ldx
    capturing = torch.cuda.is_current_stream_capturing()
    key = (x.data_ptr(), n, F)
    hit = None if capturing else _RELAID.get(key)
    if hit is not None:
        storage_ref, version, wide = hit
        if storage_ref() is not None and version == x._version and _relaid_copy_is_current(x, wide, F):
            if not dry_run:
                RELAY_STATS["hits"] = RELAY_STATS.get("hits", 0) + 1
            return wide[:, :F], F + 32
        _RELAID.pop(key, None)
        if storage_ref() is not None and version == x._version:
            RELAY_STATS["stale_copies_caught"] = RELAY_STATS.get("stale_copies_caught", 0) + 1
        del wide, hit
    free, _ = torch.cuda.mem_get_info()          # (legal during a capture: no stream operation)
    if need > free // 2:
        return x, ldx
    if dry_run:                      # describe: which kernel WOULD run (no copy is made)
        return x, F + 32
    wide = torch.empty((n, F + 32), dtype=torch.float32, device=x.device)
    L.check(L.require_gpu().tfgx_gather_rows_f32(L.ptr(x), ldx, None, n, F, L.ptr(wide), F + 32, L.stream_ptr()),
            "tfgx_gather_rows_f32 (strided row copy)")
    RELAY_STATS["copies"] += 1
    if not capturing:
        while len(_RELAID) >= 2:                 # the two most recent tables
            _RELAID.pop(next(iter(_RELAID)))
        _RELAID[key] = (_weakref.ref(x.untyped_storage()), x._version, wide)
    return wide[:, :F], F + 32


RELAY_STATS = {"copies": 0}
_RELAID = {}      # (data_ptr, n, F) -> (weakref to the table's storage, its version counter at copy time, the re-laid copy)


def _relaid_copy_is_current(x, wide, F):
    """Sampled rows of the memoised copy against x, bit for bit (the promoted layouts' check, tfgx_split_rows_verify_f32, on
    the two column halves of the copy)."""
    samples = _verify_rows_setting()
    if samples == 0:
        return True
    n = int(x.shape[0])
    if samples < 0:
        samples = n
    half = (F // 2) // 4 * 4
    lib = L.require_gpu()
    flag = torch.empty(1, dtype=torch.int32, device=x.device)
    _VERIFY_CALLS[0] += 1
    L.check(lib.tfgx_split_rows_verify_f32(L.ptr(x), int(x.stride(0)), n, F, half, L.ptr(wide), int(wide.stride(0)),
                                           wide.data_ptr() + 4 * half, int(wide.stride(0)), samples,
                                           (0x5DEECE66D * _VERIFY_CALLS[0] + 11) & 0xFFFFFFFFFFFFFFFF, L.ptr(flag), L.stream_ptr()),
            "tfgx_split_rows_verify_f32 (re-laid copy)")
    return int(flag.item()) == 0


def release_relaid_copies():
    """Drop the memoised re-laid tables (relaid_for_gather)."""
    _RELAID.clear()


def wide_blocks_hint(explicit_spans, has_hub_lists, ldx, num_edges, n_rows):
    """tfgx_reduce_args.wide_blocks for one launch: 0 = the kernel's own choice (wide line-aligned rows gathered in 64-column
    blocks), -1 = one burst per gathered row.  ONE policy for every caller that fills a ReduceArgs
    (plan.segment_reduce and the sharded backend, dist/sharded.py — ADVICE r5: the sharded passes used to miss it):
      * explicit spans (a row's own-source edges, then one sub-span per halo round: a handful of edges per row and pass) and
        short rows (fewer than 32 edges per launched row on average): a row's start-up — header loads, the first index batch,
        the self-loop row — is paid once per column pass.  One shard of the papers100M-shaped graph (13.9 M rows x 14.4 edges,
        F = 128, 57 GB table): 19.2 ms with one burst per row against 22.2 with two column passes (profiles/r05_papers_shard.jsonl)
      * a power-law plan (hub lists) whose row stride is not a power of two — the table's own, or after relaid_for_gather moved
        its rows 128 bytes further apart: same-box A/B on the products-sized R-MAT graph
        (profiles/r05_ab_wide_blocks_modes_rmat.jsonl): F = 192 / 224 13.4 / 15.7 ms with bursts against 14.6 / 18.0 with column
        blocks (mostly short rows, and the hot source rows already hit in the caches); at F = 128 / 512, where a power-of-two
        stride folds the hot rows onto few cache sets, the blocks win (11.2 -> 9.4 ms, 50.8 -> 48.1) and stay on."""
    if explicit_spans:
        return -1
    if not has_hub_lists and num_edges < 32 * max(int(n_rows), 1):
        return -1
    if has_hub_lists and not pow2_row_stride(ldx):
        return -1
    return 0


def segment_reduce(plan, x, op, w_csr=None, out=None, act=L.ACT_NONE, self_coef=None, bias=None, add_x=None,
                   accumulate=False, mean_count=None, row_begin=None, row_end=None, rp_stride=1, col=None,
                   n_dst=None, describe=False, track=None, track_row_begin=None, wide_blocks=None):
    """One launch of tfgx_segment_reduce_f32 on `plan` (or on explicit row_begin/row_end/col views of it).
    `x` is a dense [n_src, F] tensor or a SplitRows.  describe=True launches nothing and returns the kernel symbol the
    dispatcher picks for these arguments (tfgx_segment_reduce_describe).  `track` (TFGX_MAX, training forward): int32
    [n_dst, F] that receives tie count << 16 | row-relative position of the first maximal edge (tfgx_reduce_args.track;
    see can_track); with accumulate=True the launch merges into the (out, track) earlier launches stored for earlier
    sub-spans of the same rows (track_row_begin: the first position of the whole row)."""
    lib = L.require_gpu()
    split = x if isinstance(x, SplitRows) else None
    if split is not None:
        x, ldx = L.row_major_2d(split.main)
        F = split.shape[1]
    else:
        x, ldx = L.row_major_2d(x)
        F = int(x.shape[1])
        if row_begin is None and row_end is None and col is None:
            x, ldx = relaid_for_gather(x, ldx, plan.hub_info() is not None, dry_run=describe)
    n_dst = plan.n_dst if n_dst is None else int(n_dst)
    given = out is not None
    if out is None:
        out = torch.empty((n_dst, F), dtype=torch.float32, device=x.device)
    out2, ldo = L.row_major_2d(out)
    assert out2 is out, "out must be row-major"
    a = L.ReduceArgs()
    rb = plan.row_ptr if row_begin is None else row_begin
    re = plan.row_ptr[1:] if row_end is None else row_end
    a.row_begin = rb.data_ptr()
    a.row_end = re.data_ptr()
    a.rp_stride = rp_stride
    c = plan.col if col is None else col
    a.col = c.data_ptr()
    a.w = 0 if w_csr is None else w_csr.data_ptr()
    a.n_dst = n_dst
    a.x = x.data_ptr()
    a.ldx = ldx
    a.F = F
    a.out = out.data_ptr()
    a.ldo = ldo
    a.op = op
    a.act = act
    a.accumulate = 1 if accumulate else 0
    a.self_coef = 0 if self_coef is None else self_coef.data_ptr()
    a.bias = 0 if bias is None else bias.data_ptr()
    if add_x is not None:
        add_x, ld_add = L.row_major_2d(add_x)
        a.add_x = add_x.data_ptr()
        a.ld_add = ld_add
    a.mean_count = 0 if mean_count is None else mean_count.data_ptr()
    if split is not None:
        a.x_tail, a.ld_tail, a.f_main = split.tail.data_ptr(), int(split.tail.shape[1]), int(split.main.shape[1])
        if split.edge_tail is not None and split.edge_plan is plan and col is None:
            a.edge_tail, a.ld_edge_tail = split.edge_tail.data_ptr(), int(split.edge_tail.shape[1])
    if row_begin is None and row_end is None and col is None and n_dst == plan.n_dst and USE_ROW_ORDER and F <= ROW_ORDER_MAX_F:
        # skewed plans: degree-ordered walk (the rows sharing a wave have similar lengths).  Same-box A/B on R-MAT graphs:
        # F = 20: -7 .. -16 %, F = 64: -4 .. -11 % (round 2); round 4, after the hub finalize stopped being a latency chain
        # (profiles/r04_ab_row_order.jsonl, 2.4 M / 123 M and 2^21 / 61 M edges): F = 100: -10 / -11 %, F = 128: -3 / -5 %,
        # F = 256: -2 .. +1 % (one row per wave there: nothing to balance) — hence the width limit
        order = plan.row_order()
        if order is not None:
            a.row_order = order.data_ptr()
    if track is not None:
        a.track, a.ld_track = track.data_ptr(), int(track.stride(0))
        if track_row_begin is not None:      # positions relative to the WHOLE row when it is reduced span by span
            a.track_row_begin = track_row_begin.data_ptr()
    hub = plan.hub_info() if (row_begin is None and row_end is None and col is None and track is None) else None
    if hub is not None:
        hub_rows, chunk_ptr, chunk_begin, chunk_end, _ = hub
        scratch = torch.empty((int(chunk_begin.shape[0]), F), dtype=torch.float32, device=x.device)
        a.hub_threshold = plan.hub_threshold
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), chunk_ptr.data_ptr()
        a.hub_chunk_begin, a.hub_chunk_end = chunk_begin.data_ptr(), chunk_end.data_ptr()
        a.n_hub_rows, a.n_hub_chunks = int(hub_rows.shape[0]), int(chunk_begin.shape[0])
        a.hub_scratch = scratch.data_ptr()
    a.wide_blocks = wide_blocks_hint(explicit_spans=row_begin is not None or row_end is not None or col is not None,
                                     has_hub_lists=hub is not None, ldx=ldx, num_edges=plan.num_edges, n_rows=n_dst)
    if wide_blocks is not None:          # tests / A-B tools: +1 column blocks wherever the layout allows, -1 one burst per row
        a.wide_blocks = int(wide_blocks)
    if describe:
        buf = ctypes.create_string_buffer(160)
        L.check(lib.tfgx_segment_reduce_describe(ctypes.byref(a), buf, 160), "tfgx_segment_reduce_describe")
        return buf.value.decode()
    L.check(lib.tfgx_segment_reduce_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_segment_reduce_f32")
    _AGG_LAUNCHES[0] += 1
    if given:
        written_by_kernel(out)
    if track is not None:
        written_by_kernel(track)
    return out


FUSE_AGGREGATE_GEMM = True      # developer A/B switch (tools/ab_fused_layer.py): False = always two launches
TFGX_FUSE_WIDE = False          # developer A/B switch: True = the fused launch also at F = 128 on large dense tables (round 4's route)


FUSED_STATS = {"launches": 0, "with_side_output": 0}      # diagnostics (tests assert the route taken)


def aggregate_gemm_applies(x, kernel, op=L.SUM):
    """Would aggregate_gemm take this call?  (No launch; callers that must choose their autograd route ask first.)
    x: a dense table or a SplitRows (the static feature layout)."""
    if not FUSE_AGGREGATE_GEMM or op not in (L.SUM, L.MEAN) or kernel is None:
        return False
    lib = L.require_gpu()
    x2, ldx = L.row_major_2d(x.main if isinstance(x, SplitRows) else x)
    F, N = (x.shape[1] if isinstance(x, SplitRows) else int(x2.shape[1])), int(kernel.shape[1])
    return bool(int(kernel.shape[0]) == F and lib.tfgx_aggregate_gemm_fits(F, N) and ldx % 4 == 0
                and x2.data_ptr() % 16 == 0)


def aggregate_gemm(plan, x, op, kernel, w_csr=None, self_coef=None, bias=None, act=L.ACT_NONE, out=None, agg_out=None):
    """act(segment_reduce(plan, x, op, w_csr, self_coef) @ kernel + bias) in ONE launch (tfgx_aggregate_gemm_f32: the
    aggregate goes registers -> LDS -> MFMA and is never read back from HBM), or None when the fused kernel does not take
    this call (shape outside tfgx_aggregate_gemm_fits, unaligned rows) — the caller then runs the two launches.
    agg_out: optional dense [n_dst, F] tensor that ALSO receives the aggregate itself (the training forward: the weight
    gradient needs it)."""
    if not FUSE_AGGREGATE_GEMM or op not in (L.SUM, L.MEAN):
        return None
    lib = L.require_gpu()
    split = x if isinstance(x, SplitRows) else None       # static feature layout: main / tail / per-edge tail (as segment_reduce)
    x2, ldx = L.row_major_2d(split.main if split is not None else x)
    k2, ldb = L.row_major_2d(L.as_f32(kernel))
    F, N = (split.shape[1] if split is not None else int(x2.shape[1])), int(k2.shape[1])
    if int(k2.shape[0]) != F or not lib.tfgx_aggregate_gemm_fits(F, N) or ldx % 4 != 0 or x2.data_ptr() % 16 != 0:
        return None
    hub = plan.hub_info()      # long rows: chunk partials by a launch of the ordinary kernel, folded by the row's lane group
    order = plan.row_order()   # skewed plans: tiles of similar-length rows (degree order), results unchanged
    if (split is None and agg_out is None and hub is None and F >= 128 and F % 32 == 0 and ldx % 32 == 0 and x2.data_ptr() % 128 == 0
            and plan.num_edges >= 32 * max(plan.n_dst, 1) and 4 * int(x2.shape[0]) * F > (512 << 20) and not TFGX_FUSE_WIDE):
        # round 5: at F = 128 the stand-alone aggregation gathers in column blocks (10.2 -> 9.2 ms at products shape), which this
        # launch's producers do not; on a large dense table the two launches are now ahead at inference — same-box A/B
        # (profiles/r05_ab_fused_layer_F128.json): GCN layer 128 -> 256 11.05 fused vs 10.53, mean GraphSAGE 10.86 vs 10.40.
        # The TRAINING forward keeps the fused launch (its side output saves the aggregate's read-back: 14.66 vs 15.14), and so
        # do small graphs (arxiv shape: the table is cache-resident, 0.238 vs 0.259 ms).
        return None
    if split is None:
        x2, ldx = relaid_for_gather(x2, ldx, hub is not None)      # F = 128 at a 512-byte row stride on a power-law plan
    n_dst = plan.n_dst
    given = out is not None
    if out is None:
        out = torch.empty((n_dst, N), dtype=torch.float32, device=x2.device)
    _, ldc = L.row_major_2d(out)
    a = L.ReduceArgs()
    a.row_begin, a.row_end, a.rp_stride = plan.row_ptr.data_ptr(), plan.row_ptr[1:].data_ptr(), 1
    a.col = plan.col.data_ptr()
    a.w = 0 if w_csr is None else w_csr.data_ptr()
    a.n_dst, a.x, a.ldx, a.F = n_dst, x2.data_ptr(), ldx, F
    a.op = op
    a.self_coef = 0 if self_coef is None else self_coef.data_ptr()
    if split is not None:
        a.x_tail, a.ld_tail, a.f_main = split.tail.data_ptr(), int(split.tail.shape[1]), int(split.main.shape[1])
        if split.edge_tail is not None and split.edge_plan is plan:
            a.edge_tail, a.ld_edge_tail = split.edge_tail.data_ptr(), int(split.edge_tail.shape[1])
    if order is not None:
        a.row_order = order.data_ptr()
    if hub is not None:
        hub_rows, chunk_ptr, chunk_begin, chunk_end, _ = hub
        scratch = torch.empty((int(chunk_begin.shape[0]), F), dtype=torch.float32, device=x2.device)
        a.hub_threshold = plan.hub_threshold
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), chunk_ptr.data_ptr()
        a.hub_chunk_begin, a.hub_chunk_end = chunk_begin.data_ptr(), chunk_end.data_ptr()
        a.n_hub_rows, a.n_hub_chunks = int(hub_rows.shape[0]), int(chunk_begin.shape[0])
        a.hub_scratch = scratch.data_ptr()
        slot = plan.hub_order_slot()
        if slot is not None:
            a.hub_order_slot = slot.data_ptr()
    if agg_out is not None:
        ao, ldo = L.row_major_2d(agg_out)
        assert ao is agg_out and tuple(agg_out.shape) == (n_dst, F) and ldo % 4 == 0 and agg_out.data_ptr() % 16 == 0
        a.out, a.ldo = agg_out.data_ptr(), ldo
        FUSED_STATS["with_side_output"] += 1
    FUSED_STATS["launches"] += 1
    _AGG_LAUNCHES[0] += 1
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    L.check(lib.tfgx_aggregate_gemm_f32(ctypes.byref(a), L.ptr(k2), ldb, L.ptr(bias_t), act, L.ptr(out), ldc, N,
                                        L.stream_ptr()), "tfgx_aggregate_gemm_f32")
    if given:
        written_by_kernel(out)
    written_by_kernel(agg_out)
    return out


def can_track(plan, x2, ldx):
    """Can the TRAINING forward of max aggregation run on the tuned segment-reduce kernel with packed tie counts / first
    positions (tfgx_reduce_args.track)?  16-byte aligned rows of 32 <= F <= 256 columns — or wider line-aligned rows — (the
    dwordx4, one-chunk-per-lane instantiations), no hub rows (hub rows are chunked and do not track) and every row shorter than 65536 edges."""
    F = int(x2.shape[1])
    # one 16-byte chunk per lane: rows up to 256 columns, or — round 5 — any wider row made of whole 128-byte lines, which the
    # kernel walks in 64-column blocks on grid.y (tfgx_reduce.hip group_shape; TFGX_REDUCE_WIDE_BLOCKS=0 switches them off)
    wide = (F % 32 == 0 and ldx % 32 == 0 and x2.data_ptr() % 128 == 0 and _os.environ.get("TFGX_REDUCE_WIDE_BLOCKS", "1") != "0"
            and plan.num_edges >= 32 * max(plan.n_dst, 1))          # (segment_reduce runs sparse plans as one burst per row)
    return (F % 4 == 0 and F >= 32 and (F <= 256 or wide) and ldx % 4 == 0 and x2.data_ptr() % 16 == 0 and plan.hub_info() is None
            and int(getattr(plan, "hub_threshold", 1 << 30)) < 65536)


def _avg_lines(F, ld):
    """Average number of 128-byte lines a row of F floats touches when rows start ld floats apart (rows 4-byte aligned)."""
    import math
    row, stride = 4 * F, 4 * ld
    period = 128 // math.gcd(stride, 128)
    return sum(((i * stride) % 128 + row - 1) // 128 + 1 for i in range(period)) / period


def gather_friendly_ld(F):
    """Row stride (in floats) that keeps a GATHERED [*, F] row on the fewest 128-byte lines.  The aggregation kernels run
    at the part's line-request ceiling, so a row that straddles an extra line costs exactly that much.  Candidates: F
    itself, the next power of two (F <= 32), the next multiple of 16 and of 32 floats; the one with the fewest lines per
    row on average wins, ties go to the smaller stride.  20 -> 32 (1.5 -> 1 line: 3.32 -> 2.36 ms per products-shaped
    pass), 47 -> 48 (2.44 -> 2: 5.14 -> 4.35 ms), 44 -> 48 (2.25 -> 2), 172 -> 176 (6.25 -> 6); 40, 64, 100 stay
    dense; 128 / 256 / 512 ... get one more line per row (160 / 288 / 544: pow2_row_stride) (F = 100 touches 4 lines at any stride; padding it to 112 / 128 LOSES 3-9 % to the larger footprint).
    Same-box A/B: tools/ab_row_stride.py, profiles/r02_ab_row_stride.jsonl."""
    F = int(F)
    if F <= 0:
        return max(F, 1)
    cands = [F, (F + 15) // 16 * 16, (F + 31) // 32 * 32]
    if F <= 32:
        cands.append(1 << (F - 1).bit_length())
    best = min(sorted(set(cands)), key=lambda ld: (round(_avg_lines(F, ld), 6), ld))
    if pow2_row_stride(best):
        best += 32          # never a power-of-two row stride of 512 bytes or more: see pow2_row_stride
    return best


def pow2_row_stride(ld):
    """Is a row stride of `ld` floats a power of two of at least 512 bytes?  Rows that far apart put the same column of every
    row on the same few memory channels / cache sets.  Round 5, same box, products-sized graphs (profiles/r05_ab_ld_pad.jsonl:
    ld = F against ld = F + 32): on the R-MAT graph, whose hot source rows are what the aliasing folds together, F = 128
    10.2 -> 7.7 ms, F = 256 25.6 -> 19.3, F = 512 49.3 -> 40.0; on the uniform graph F = 512 43.8 -> 36.7 ms (F = 128 / 256:
    9.20 -> 9.03 / 20.8 -> 20.9).  L2 hit rates barely move (17.7 -> 19.9 % at F = 256, profiles/r05_rmat_stride_pmc.md): the
    conflict is beyond L2.  Tables this package lays out avoid such strides (gather_friendly_ld); a caller's table that has
    one is re-laid out on the fly where that pays (relaid_for_gather)."""
    ld = int(ld)
    return ld >= 128 and (ld & (ld - 1)) == 0


def gather_friendly_empty(n, F, device):
    """Uninitialised float32 [n, F] whose rows start gather_friendly_ld(F) floats apart (a column view of a slightly
    wider buffer when that differs from F).  For tensors this package PRODUCES and then gathers by row — GEMM outputs that
    feed an aggregation, gradients fed to the transposed aggregation — never for the caller's own arrays."""
    ld = gather_friendly_ld(F)
    buf = torch.empty((int(n), ld), dtype=torch.float32, device=device)
    return buf if ld == int(F) else buf[:, :int(F)]


def gather_friendly_copy(t):
    """`t` itself when its rows already sit on a gather-friendly stride, else a copy that does."""
    F = int(t.shape[1])
    if t.stride(1) == 1 and t.stride(0) == gather_friendly_ld(F):
        return t
    if gather_friendly_ld(F) == F:
        return t.contiguous()
    out = gather_friendly_empty(int(t.shape[0]), F, t.device)
    out.copy_(t)
    return out


def gemm_bias_act(a, b, bias=None, act=L.ACT_NONE, out=None, act_cols=None):
    """act(a @ b + bias) on the fp32 MFMA kernel; `act_cols`: activate only columns [0, act_cols)."""
    lib = L.require_gpu()
    a, lda = L.row_major_2d(L.as_f32(a))
    b, ldb = L.row_major_2d(L.as_f32(b))
    M, K = int(a.shape[0]), int(a.shape[1])
    if int(b.shape[0]) != K:
        raise ValueError("matmul shape mismatch: {} @ {}".format(tuple(a.shape), tuple(b.shape)))
    N = int(b.shape[1])
    given = out is not None
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _, ldc = L.row_major_2d(out)
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    ws_bytes = lib.tfgx_gemm_workspace_bytes(M, K, N)       # split-K partials (small M, long K) or the row kernel's tile counters (tall M)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
    L.check(lib.tfgx_gemm_bias_act_cols_ws_f32(L.ptr(a), lda, L.ptr(b), ldb, L.ptr(bias_t), act,
                                               N if act_cols is None else int(act_cols), L.ptr(out), ldc, M, K, N,
                                               L.ptr(ws), ws_bytes, L.stream_ptr()), "tfgx_gemm_bias_act_cols_ws_f32")
    if given:
        written_by_kernel(out)
    return out


def l2_normalize_rows_(h):
    lib = L.require_gpu()
    h2, ld = L.row_major_2d(h)
    assert h2 is h
    L.check(lib.tfgx_l2_normalize_rows_f32(L.ptr(h), ld, int(h.shape[0]), int(h.shape[1]), L.stream_ptr()),
            "tfgx_l2_normalize_rows_f32")
    return h


def gather_rows(x, idx, out=None):
    lib = L.require_gpu()
    x, ldx = L.row_major_2d(x)
    idx = L.as_i32(idx)
    M, F = int(idx.shape[0]), int(x.shape[1])
    given = out is not None
    if out is None:
        out = torch.empty((M, F), dtype=torch.float32, device=x.device)
    _, ldo = L.row_major_2d(out)
    L.check(lib.tfgx_gather_rows_f32(L.ptr(x), ldx, L.ptr(idx), M, F, L.ptr(out), ldo, L.stream_ptr()),
            "tfgx_gather_rows_f32")
    if given:
        written_by_kernel(out)
    return out


def gemm_tn(x, g, want_bias=False, gate=None):
    """(x^T @ g, column sums of g or None): the weight / bias gradient of a dense layer on the MFMA reduction kernel
    (tfgx_gemm_tn_f32).  x [M, Ka], g [M, N] -> dW [Ka, N], db [N].  `gate` [M, N]: count g only where gate > 0 (the
    ReLU of the layer's epilogue, gate = the layer's output) — the masked gradient is never materialised."""
    lib = L.require_gpu()
    x, ldx = L.row_major_2d(L.as_f32(x))
    g, ldg = L.row_major_2d(L.as_f32(g))
    M, Ka, N = int(x.shape[0]), int(x.shape[1]), int(g.shape[1])
    if int(g.shape[0]) != M:
        raise ValueError("gemm_tn: x has {} rows, g has {}".format(M, int(g.shape[0])))
    ldt = 0
    if gate is not None:
        gate, ldt = L.row_major_2d(gate)
        if tuple(gate.shape) != (M, N):
            raise ValueError("gemm_tn: gate has shape {}, g has {}".format(tuple(gate.shape), (M, N)))
    dW = torch.empty((Ka, N), dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = lib.tfgx_gemm_tn_workspace_bytes(M, Ka, N, 1 if want_bias else 0)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x.device)
    L.check(lib.tfgx_gemm_tn_gated_f32(L.ptr(x), ldx, L.ptr(g), ldg, L.ptr(gate), ldt, M, Ka, N, L.ptr(dW), N, L.ptr(db),
                                       L.ptr(ws), ws_bytes, L.stream_ptr()), "tfgx_gemm_tn_gated_f32")
    return dW, db


def column_sums(g):
    """g.sum(0) of a [M, N] float32 matrix on the two-phase kernel (tfgx_column_sum_f32): deterministic, and at HBM rate for
    every width (torch's reduction runs a [2.4 M, 47] gradient — ogbn-products' 47 classes — at 24 GB/s: 18.7 ms per step)."""
    lib = L.require_gpu()
    g2, ldg = L.row_major_2d(L.as_f32(g))
    M, N = int(g2.shape[0]), int(g2.shape[1])
    out = torch.empty(N, dtype=torch.float32, device=g2.device)
    ws_bytes = lib.tfgx_column_sum_workspace_bytes(M, N)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=g2.device)
    L.check(lib.tfgx_column_sum_f32(L.ptr(g2), ldg, M, N, L.ptr(out), L.ptr(ws), ws_bytes, L.stream_ptr()), "tfgx_column_sum_f32")
    return out


def transpose(a):
    """a^T as a new row-major tensor (tfgx_transpose_f32)."""
    lib = L.require_gpu()
    a, lda = L.row_major_2d(L.as_f32(a))
    r, c = int(a.shape[0]), int(a.shape[1])
    out = torch.empty((c, r), dtype=torch.float32, device=a.device)
    L.check(lib.tfgx_transpose_f32(L.ptr(a), lda, r, c, L.ptr(out), r, L.stream_ptr()), "tfgx_transpose_f32")
    return out
