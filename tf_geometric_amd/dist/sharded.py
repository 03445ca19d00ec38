# coding=utf-8
"""Destination-range sharding of ONE large graph over the GPUs of a node, with halo exchange.

The reference has nothing like this: its two "distributed" demos replicate the whole graph on every GPU and only
all-reduce gradients (demo/demo_distributed_gcn.py:38-57).  Here aggregation is sharded where it is independent —
per destination node:

  * rank g owns a contiguous destination range [lo_g, hi_g), split points chosen on the global row_ptr so that
    EDGES (not nodes) are balanced; it owns those rows of x, their CSR rows and the output rows;
  * a rank's sources are its own rows plus a "halo": the sorted, de-duplicated remote rows its edges reference.
    The local source table is [own rows | halo rows], and col is remapped into it once (plan time);
  * per layer ONE all-to-all-v moves halo rows (RCCL over xGMI: every GPU pair has its own link, so the
    personalised exchange uses all 7 links at once — unlike a ring all-reduce);
  * the exchange runs asynchronously while the LOCAL-source edges are aggregated; the HALO-source edges are then
    accumulated into the same output rows (tfgx_reduce_args.accumulate) and the epilogue (self-loop term, mean
    divisor, bias, activation) is applied once, in that second pass.

Compute goes through a backend object whose methods map 1:1 onto C-ABI entry points (HipBackend, the default
and the only one in this package).  tests/ injects a numpy backend to exercise THIS file's orchestration with
world_size-2 gloo process groups on CPU; the product path never does.
"""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib as L


class HipBackend(object):
    """Thin adapter: every method is one (or a fixed short sequence of) C-ABI call(s) on the current HIP stream."""
    name = "hip"

    def __init__(self):
        self.lib = L.require_gpu()
        self.device = L.device()

    # ---- memory
    def i32(self, a):
        return L.as_i32(a, self.device)

    def f32(self, a):
        return L.as_f32(a, self.device)

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ---- plan
    def build_csr(self, edge_index, n_dst, n_src):
        from ..plan import CsrPlan
        p = CsrPlan.build(edge_index, n_dst, n_src)
        return p.row_ptr, p.col, p.perm

    def permute_rows(self, attr, perm):
        a = self.f32(attr).contiguous()
        out = torch.empty_like(a)
        width = 1 if a.dim() == 1 else int(a.shape[1])
        L.check(self.lib.tfgx_permute_rows_f32(L.ptr(a), L.ptr(perm), int(perm.shape[0]), width, L.ptr(out),
                                               L.stream_ptr()), "tfgx_permute_rows_f32")
        return out

    def halo_plan(self, col, src_lo, src_hi, n_global, n_own, peer_bounds=None, rank=0, dense_pct=0):
        """-> (halo_ids sorted int32 [n_halo], col_local int32 [E]).  Sources in [src_lo, src_hi) are resident own rows
        (local index c - src_lo); every other source becomes a halo row at table index n_own + its rank among the halo ids.
        `peer_bounds` + `dense_pct`: a peer whose block is referenced to at least dense_pct percent is requested WHOLE
        (flags of its range set before the compaction), so that its owner can send the block without packing."""
        E = int(col.shape[0])
        flags = self.empty(max(n_global, 1), torch.int32)
        pos = self.empty(max(n_global, 1), torch.int32)
        ids = self.empty(max(n_global, 1), torch.int32)
        n_halo = torch.zeros(1, dtype=torch.int32, device=self.device)
        ws_bytes = self.lib.tfgx_halo_workspace_bytes(n_global)
        ws = self.empty(max(ws_bytes, 1), torch.uint8)
        L.check(self.lib.tfgx_halo_mark(L.ptr(col), E, src_lo, src_hi, n_global, L.ptr(flags), L.stream_ptr()),
                "tfgx_halo_mark")
        if peer_bounds is not None and dense_pct > 0 and n_global > 0:
            _fill_dense_peers(flags[:n_global], peer_bounds, rank, dense_pct)
        L.check(self.lib.tfgx_halo_compact(L.ptr(flags), n_global, L.ptr(pos), L.ptr(ids), L.ptr(n_halo), L.ptr(ws),
                                           ws_bytes, L.stream_ptr()), "tfgx_halo_compact")
        col_local = self.empty(E, torch.int32)
        L.check(self.lib.tfgx_halo_remap_cols(L.ptr(col), E, src_lo, src_hi, L.ptr(pos), n_own,
                                              L.ptr(col_local), L.stream_ptr()), "tfgx_halo_remap_cols")
        k = int(n_halo.item())
        return ids[:k].clone(), col_local

    def split_by_class(self, row_ptr, col_local, w, class_bounds, n_class):
        n_dst, E = int(row_ptr.shape[0]) - 1, int(col_local.shape[0])
        rpk = self.empty(n_class * n_dst + 1, torch.int32)
        col2 = torch.empty_like(col_local)
        w2 = None if w is None else torch.empty_like(w)
        bounds = self.i32(np.asarray(list(class_bounds) + [2 ** 31 - 1], dtype=np.int64))
        L.check(self.lib.tfgx_split_by_source_class(L.ptr(row_ptr), L.ptr(col_local), L.ptr(w), n_dst, E, L.ptr(bounds),
                                                    n_class, L.ptr(rpk), L.ptr(col2), L.ptr(w2), L.stream_ptr()),
                "tfgx_split_by_source_class")
        return rpk, col2, w2

    # ---- compute
    def gather_rows(self, x, idx, out=None):
        from ..plan import gather_rows
        return gather_rows(x, idx, out=out)

    def scatter_add_rows(self, dst, idx, src):
        """dst[idx[i]] += src[i] with unique idx (tfgx_scatter_add_rows_f32)."""
        _, ldd = L.row_major_2d(dst)
        src, lds = L.row_major_2d(src)
        L.check(self.lib.tfgx_scatter_add_rows_f32(L.ptr(dst), ldd, L.ptr(idx), int(idx.shape[0]), int(dst.shape[1]),
                                                   L.ptr(src), lds, L.stream_ptr()), "tfgx_scatter_add_rows_f32")
        return dst

    def linear(self, x, kernel, bias=None, act=L.ACT_NONE):
        """Differentiable act(x @ kernel + bias): forward AND backward on the MFMA kernels (autograd.linear)."""
        from .. import autograd as AG
        return AG.linear(x, kernel, bias, act)

    def aggregate_autograd(self, sg, table, op, w, handles=None):
        """Differentiable reduce of w * table[col] over the shard's rows, table = [own | halo] (n_own x n_table
        operator): the single-GPU autograd functions on the shard's rectangular plan.  `handles`: a halo exchange still
        in flight for `table` (max only) — the forward then runs SPAN BY SPAN, own-source edges under the whole exchange
        and the sub-span of round j as soon as round j has landed, merging maxima / tie counts / first positions in the
        kernel epilogue (tfgx_reduce_args.track with accumulate); the backward is the single-GPU one either way."""
        from .. import autograd as AG
        passes = sg.tracked_max_passes(table, handles) if (op == L.MAX and handles is not None) else None
        if handles is not None and passes is None:
            with torch.no_grad():
                sg.exchange_finish(handles)
        return AG.aggregate(sg.local_plan(), table, op, w_csr=w, max_passes=passes)

    def gat_attention_autograd(self, sg, Q, K, V, num_heads, handles=None):
        """Differentiable fused attention of own destinations over table sources (self-loop = table row r, own rows
        come first in the table).  `handles`: the exchange of the [K | V] table still in flight — the forward then runs as
        the inference path does (own-source span under the exchange, halo span after it, states merged; the merge also
        writes the softmax statistics the backward reads); the backward is the single-GPU one."""
        from .. import autograd as AG
        passes = None
        if handles is not None:
            def passes(Qd, Kd, Vd, stats):
                with torch.no_grad():
                    return sg._gat_attention_spans(Qd, Kd, Vd, num_heads, handles, None, L.ACT_NONE, None, stats=stats)
            base = getattr(K, "_base", None)
            if base is not None and getattr(V, "_base", None) is base and K.data_ptr() == base.data_ptr():
                # K and V are the column blocks of ONE halo table: the backward writes d[K | V] into one buffer and sends
                # its halo rows back before it computes the own rows (matched to the table's exchange node by its storage)
                token = int(base.data_ptr())
                passes.halo_first = (sg.n_own, sg.halo_round_windows(),
                                     sg._early_reverse_rounds(token, "gat_halo_first_backwards", check_ptr=False))
        return AG.gat_attention(sg.local_plan(), Q, K, V, num_heads, passes=passes)

    def hub_lists(self, row_begin, row_end, rp_stride, n_dst, num_edges):
        """Chunk lists for the long spans of one pass (skewed graphs), or None — see plan.hub_policy."""
        from ..plan import build_hub_lists, hub_policy
        thr, chunk = hub_policy(num_edges, n_dst)
        idx = torch.arange(n_dst, device=row_begin.device) * rp_stride
        lists = build_hub_lists(row_begin[idx], row_end[idx], thr, chunk)
        return None if lists is None else (thr,) + lists

    def split_rows(self, table, col):
        """Static source table -> (main, tail, edge_tail): the edge-resident-tail layout of DESIGN.md §2.1 for a shard's
        [own | halo] table, edge_tail in THIS shard's CSR order; None when the width does not call for it."""
        from ..plan import SplitRows, gather_rows
        n, F = int(table.shape[0]), int(table.shape[1])
        if not SplitRows.wanted(n, F):
            return None
        sp = SplitRows.from_dense(table)
        return sp.main, sp.tail, gather_rows(sp.tail, col)

    def segment_reduce(self, row_begin, row_end, rp_stride, col, w, n_dst, x, out, op, act=L.ACT_NONE,
                       accumulate=False, self_coef=None, bias=None, mean_count=None, hub=None, split=None, track=None,
                       track_row_begin=None, describe=False):
        F_total = int(x.shape[1])
        if split is not None:
            x = split[0]
        x, ldx = L.row_major_2d(x)
        _, ldo = L.row_major_2d(out)
        a = L.ReduceArgs()
        a.row_begin, a.row_end, a.rp_stride = row_begin.data_ptr(), row_end.data_ptr(), rp_stride
        a.col = col.data_ptr()
        a.w = 0 if w is None else w.data_ptr()
        a.n_dst = n_dst
        a.x, a.ldx, a.F = x.data_ptr(), ldx, F_total
        if split is not None:
            main, tail, edge_tail = split
            a.x_tail, a.ld_tail, a.f_main = tail.data_ptr(), int(tail.shape[1]), int(main.shape[1])
            if hub is None:       # chunk scratch passes re-walk spans with their own positions: keep those dense-tail
                a.edge_tail, a.ld_edge_tail = edge_tail.data_ptr(), int(edge_tail.shape[1])
        a.out, a.ldo = out.data_ptr(), ldo
        a.op, a.act, a.accumulate = op, act, 1 if accumulate else 0
        a.self_coef = 0 if self_coef is None else self_coef.data_ptr()
        a.bias = 0 if bias is None else bias.data_ptr()
        a.mean_count = 0 if mean_count is None else mean_count.data_ptr()
        if track is not None:
            a.track, a.ld_track = track.data_ptr(), int(track.stride(0))
            a.track_row_begin = 0 if track_row_begin is None else track_row_begin.data_ptr()
        if hub is not None:
            thr, hub_rows, chunk_ptr, chunk_begin, chunk_end, _ = hub
            scratch = self.empty((int(chunk_begin.shape[0]), F_total))
            a.hub_threshold = thr
            a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), chunk_ptr.data_ptr()
            a.hub_chunk_begin, a.hub_chunk_end = chunk_begin.data_ptr(), chunk_end.data_ptr()
            a.n_hub_rows, a.n_hub_chunks = int(hub_rows.shape[0]), int(chunk_begin.shape[0])
            a.hub_scratch = scratch.data_ptr()
        # the per-class passes of a shard are span launches (a row's own-source edges, then one sub-span per halo round): the
        # one policy of plan.wide_blocks_hint (this path used to leave the kernel's column blocks on)
        from ..plan import wide_blocks_hint
        a.wide_blocks = wide_blocks_hint(explicit_spans=rp_stride > 1, has_hub_lists=hub is not None, ldx=ldx,
                                         num_edges=int(col.shape[0]) // max(int(rp_stride), 1), n_rows=n_dst)
        if describe:
            buf = ctypes.create_string_buffer(160)
            L.check(self.lib.tfgx_segment_reduce_describe(ctypes.byref(a), buf, 160), "tfgx_segment_reduce_describe")
            return buf.value.decode()
        L.check(self.lib.tfgx_segment_reduce_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_segment_reduce_f32")
        return out

    def weight_sum(self, row_ptr, w, n, diag):
        deg = self.empty(n)
        L.check(self.lib.tfgx_segment_weight_sum_f32(L.ptr(row_ptr), L.ptr(w), n, float(diag), L.ptr(deg),
                                                     L.stream_ptr()), "tfgx_segment_weight_sum_f32")
        return deg

    def gcn_norm_edges(self, row_ptr, col, w, n, row_deg, mode, fill, add_self_loop, renorm, col_deg=None):
        E = int(col.shape[0])
        w_out, self_coef = self.empty(E), self.empty(n)
        L.check(self.lib.tfgx_gcn_norm_edges_f32(L.ptr(row_ptr), L.ptr(col), L.ptr(w), n, L.ptr(row_deg), L.ptr(col_deg), mode,
                                                 float(fill), int(add_self_loop), int(renorm), L.ptr(w_out),
                                                 L.ptr(self_coef), L.stream_ptr()), "tfgx_gcn_norm_edges_f32")
        return w_out, self_coef

    def gemm_bias_act(self, a, b, bias=None, act=L.ACT_NONE, out=None):
        from ..plan import gemm_bias_act
        return gemm_bias_act(a, b, bias=bias, act=act, out=out)

    def gat_pass(self, row_begin, row_end, rp_stride, col, n_dst, Q, K, V, num_heads, state_acc, state_ml,
                 skip_longer_than=0, part_row=None):
        """Raw online-softmax state of every launched part over its edge span (tfgx_gat_fused_f32, state mode).  A part
        is a destination row (part_row None) or a chunk of one (part_row[p] = its destination); spans longer than
        skip_longer_than (> 0) are left to the chunk launch."""
        from ..nn.conv.gat import gat_args
        a, _, keep = gat_args(Q, K, V, num_heads, n_dst, col, add_self_loop=False, out=state_acc)
        a.row_begin, a.row_end, a.rp_stride = row_begin.data_ptr(), row_end.data_ptr(), rp_stride
        a.state_acc, a.state_ml = state_acc.data_ptr(), state_ml.data_ptr()
        a.hub_threshold = int(skip_longer_than)
        if part_row is not None:
            a.hub_chunk_row = part_row.data_ptr()
        L.check(self.lib.tfgx_gat_fused_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_gat_fused_f32")

    def gat_merge_parts(self, Q, K, V, num_heads, n_dst, state_acc, state_ml, part_ptr, part_idx, bias, act, out,
                        stats=None):
        from ..nn.conv.gat import gat_args
        a, out, keep = gat_args(Q, K, V, num_heads, n_dst, self.empty(1, torch.int32), add_self_loop=True, bias=bias,
                                act=act, out=out)
        if stats is not None:          # final (m, l) per row and head: what the backward of the attention reads
            a.stats_ml = stats.data_ptr()
        L.check(self.lib.tfgx_gat_merge_parts_f32(ctypes.byref(a), L.ptr(state_acc), L.ptr(state_ml), L.ptr(part_ptr),
                                                  L.ptr(part_idx), L.stream_ptr()), "tfgx_gat_merge_parts_f32")
        return out

    def gat_merge(self, Q, K, V, num_heads, n_dst, state_acc, state_ml, n_passes, bias, act, out, stats=None):
        from ..nn.conv.gat import gat_args
        a, out, keep = gat_args(Q, K, V, num_heads, n_dst, self.empty(1, torch.int32), add_self_loop=True, bias=bias,
                                act=act, out=out)
        if stats is not None:
            a.stats_ml = stats.data_ptr()
        L.check(self.lib.tfgx_gat_merge_passes_f32(ctypes.byref(a), L.ptr(state_acc), L.ptr(state_ml), n_passes,
                                                   L.stream_ptr()), "tfgx_gat_merge_passes_f32")
        return out


def _fill_dense_peers(flags, peer_bounds, rank, dense_pct):
    """flags: int32 [n_global] (1 = referenced remote source).  Sets the whole range of every peer (not `rank`) whose
    block is referenced to at least dense_pct percent; torch ops only, so the numpy test backend shares it."""
    b = [int(v) for v in peer_bounds]
    cs = torch.zeros(flags.shape[0] + 1, dtype=torch.int64, device=flags.device)
    torch.cumsum(flags, 0, out=cs[1:])
    edges = cs[torch.as_tensor(b, dtype=torch.long, device=flags.device)].tolist()        # world + 1 integers
    for p in range(len(b) - 1):
        size, cnt = b[p + 1] - b[p], int(edges[p + 1] - edges[p])
        if p != rank and size > 0 and 100 * cnt >= dense_pct * size:
            flags[b[p]:b[p + 1]] = 1


def bounds_from_degrees(deg, world):
    """edge_balanced_bounds on a DEVICE in-degree histogram (int64 [n]): nothing node-sized visits the host.
    -> (bounds int64 numpy [world+1], row_ptr int64 tensor [n+1])."""
    n = int(deg.shape[0])
    rp = torch.zeros(n + 1, dtype=torch.int64, device=deg.device)
    torch.cumsum(deg, 0, out=rp[1:])
    E = int(rp[-1].item()) if n else 0
    bounds = np.zeros(world + 1, dtype=np.int64)
    bounds[world] = n
    if world > 1 and E > 0:
        targets = torch.tensor([(E * k) // world for k in range(1, world)], dtype=torch.int64, device=deg.device)
        bounds[1:world] = torch.searchsorted(rp, targets, right=False).cpu().numpy()
    bounds = np.minimum(np.maximum.accumulate(bounds), n)
    if E == 0:
        bounds = (np.arange(world + 1, dtype=np.int64) * n) // world
    return bounds, rp


def edge_balanced_bounds(row_ptr_host, world):
    """Split points on the global row_ptr so that each rank gets ~E/world edges. -> int64 [world+1] node ids."""
    rp = np.asarray(row_ptr_host, dtype=np.int64)
    n = rp.shape[0] - 1
    E = int(rp[-1])
    bounds = np.zeros(world + 1, dtype=np.int64)
    bounds[world] = n
    for k in range(1, world):
        bounds[k] = int(np.searchsorted(rp, (E * k) // world, side="left"))
    bounds = np.minimum(np.maximum.accumulate(bounds), n)
    if E == 0:      # no edges: balance nodes
        bounds = (np.arange(world + 1, dtype=np.int64) * n) // world
    return bounds


_DEBUG_CHECKS = os.environ.get("TFGX_DIST_DEBUG_CHECKS", "0") != "0"     # extra consistency checks that synchronise


def _bias_add(out, bias):
    """out + bias of the trainable layers: on the GPU the bias gradient is the column-sum kernel, not torch's g.sum(0)
    (autograd.bias_add); CPU tensors of the numpy test backend take the plain add."""
    if out.is_cuda:
        from .. import autograd as AG
        return AG.bias_add(out, bias)
    return out + bias


class ShardedGraph(object):
    """One rank's shard: destination rows [own_lo, own_hi) of a global graph, plus its halo plan."""

    def __init__(self):
        self.backend = None
        self.group = None
        self.rank = 0
        self.world = 1

    # ------------------------------------------------------------------ construction
    def _init_common(self, num_nodes, group, backend, transport, self_halo_rows):
        from .transport import get_transport
        be = self.backend = backend or HipBackend()
        self.group = group
        self.counters = {}      # diagnostics: how often the span-by-span training forwards ran (tests assert on them)
        inited = group is not None or dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.rank = dist.get_rank(group) if inited else 0
        self.n_global = int(num_nodes)
        self.transport = transport if (transport is not None and not isinstance(transport, str)) else \
            get_transport(group, be, transport)
        # TEST MODE (world-size-1 runs of the real exchange): only the first `self_halo_rows` own rows count as resident
        # sources; the other own rows are requested through the halo exchange — from this rank itself.  Pack kernel,
        # RCCL sends / receives, events, per-round waits and the reverse exchange then carry real rows on ONE GPU.
        self._self_halo_rows = None if self_halo_rows is None else int(self_halo_rows)
        return be

    @staticmethod
    def from_global(edge_index, num_nodes, edge_weight=None, group=None, backend=None, rounds=None, transport=None,
                    self_halo_rows=None):
        """Every rank passes the SAME global edge_index [2, E] (numpy on the host, or a tensor) and optional
        edge_weight [E]; each rank counts in-degrees over the whole list (to agree on the split points) but sorts and
        keeps only its own E/W slice.  The edge-sized work runs on the backend's device (torch ops: histogram, prefix sum,
        mask, compaction), chunk by chunk."""
        self = ShardedGraph()
        be = self._init_common(num_nodes, group, backend, transport, self_halo_rows)
        # The global list is walked in CHUNKS of 2^26 edges: a host array is copied chunk by chunk (never whole), and the
        # int64 / bool temporaries of the histogram and of the ownership mask are chunk-sized — device memory stays
        # O(E / world + chunk) instead of ~30 bytes per GLOBAL edge (papers100M: 1.6e9 edges).  from_partitioned is the
        # constructor that does not replicate the list at all.
        host = not isinstance(edge_index, torch.Tensor)
        ei_src = np.asarray(edge_index).reshape(2, -1) if host else edge_index.reshape(2, -1)
        E = int(ei_src.shape[1])
        step = int(os.environ.get("TFGX_FROM_GLOBAL_CHUNK", str(1 << 26)))
        chunks = [(c0, min(c0 + step, E)) for c0 in range(0, E, step)] or [(0, 0)]
        piece = lambda c0, c1: be.i32(np.ascontiguousarray(ei_src[:, c0:c1]) if host else ei_src[:, c0:c1]).reshape(2, -1)   # noqa: E731
        # 1. in-degree histogram of the GLOBAL edge list -> edge-balanced split points (identical on every rank)
        deg = None
        for c0, c1 in chunks:
            ei_c = piece(c0, c1)
            if ei_c.numel() and (int(ei_c.min()) < 0 or int(ei_c.max()) >= self.n_global):
                raise L.TfgxError("edge endpoint outside [0, {})".format(self.n_global))
            d = torch.bincount(ei_c[0].long(), minlength=self.n_global)
            deg = d if deg is None else deg.add_(d)
        self.bounds, rp = bounds_from_degrees(deg, self.world)
        lo, hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.own_lo, self.own_hi, self.n_own = lo, hi, hi - lo
        self.num_edges_global = int(rp[-1].item())
        # 2. my edges (destination in [lo, hi)), kept in the caller's relative order, then the stable CSR build of that
        #    slice alone — the same rows a global stable sort would put in positions [rp[lo], rp[hi])
        ids, rows, cols = [], [], []
        for c0, c1 in chunks:
            ei_c = piece(c0, c1)
            pos = torch.nonzero((ei_c[0] >= lo) & (ei_c[0] < hi)).flatten()
            ids.append(pos + c0)
            rows.append(ei_c[0][pos] - lo)
            cols.append(ei_c[1][pos])
        edge_ids = torch.cat(ids)
        local = torch.stack([torch.cat(rows), torch.cat(cols)]).to(torch.int32)
        if edge_weight is None:
            w_local = None
        elif isinstance(edge_weight, torch.Tensor):
            w_local = be.f32(edge_weight)[edge_ids]
        else:                    # host weights: only this rank's entries go to the device
            w_local = be.f32(np.asarray(edge_weight, dtype=np.float32)[edge_ids.cpu().numpy()])
        assert int(local.shape[1]) == int((rp[hi] - rp[lo]).item())
        self._finish_build(local, w_local, edge_ids, rounds)
        return self

    @staticmethod
    def from_partitioned(edge_index_part, num_nodes, edge_weight_part=None, group=None, backend=None, rounds=None,
                         transport=None, self_halo_rows=None):
        """Every rank passes ITS OWN PART of the global edge list (any split: file shards, a generator's stripes —
        destinations need not be local) as a [2, E_part] array / tensor, and optionally that part's weights.  Nothing
        edge-sized is ever replicated and nothing edge- or node-sized visits the host: the ranks all-reduce the in-degree
        histogram on the device to agree on the edge-balanced split points, bucket their edges by destination owner (stable
        sort on the device) and route them — int32 rows, int32 columns, float32 weights — with one all-to-all-v each through
        the transport; from there the build is what from_global builds.  "This shard's edge order" (self.perm, weights
        passed later) is the arrival order: parts concatenated by sending rank, each in its own order."""
        self = ShardedGraph()
        be = self._init_common(num_nodes, group, backend, transport, self_halo_rows)
        t = self.transport
        ei = be.i32(edge_index_part).reshape(2, -1)
        if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= self.n_global):
            raise L.TfgxError("edge endpoint outside [0, {})".format(self.n_global))
        w_part = None if edge_weight_part is None else be.f32(edge_weight_part).reshape(-1)
        deg = torch.bincount(ei[0].long(), minlength=self.n_global)
        if self.world > 1:
            deg = t.all_reduce_sum_i64(deg.contiguous())
        self.bounds, rp = bounds_from_degrees(deg, self.world)
        lo, hi = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.own_lo, self.own_hi, self.n_own = lo, hi, hi - lo
        self.num_edges_global = int(rp[-1].item())
        if self.world > 1:
            # destination owner of every edge of my part; a stable sort by owner keeps each part's own order
            cuts = torch.as_tensor(self.bounds[1:-1], dtype=torch.int64, device=ei.device)
            owner = torch.bucketize(ei[0].long(), cuts, right=True).clamp_(max=self.world - 1)
            order = torch.argsort(owner, stable=True)
            send_counts = torch.bincount(owner, minlength=self.world)
            send_list = [int(v) for v in send_counts.tolist()]
            recv_counts = t.all_to_all_v(send_counts.contiguous(), [1] * self.world, [1] * self.world)
            recv_list = [int(v) for v in recv_counts.tolist()]
            rows = t.all_to_all_v(ei[0][order], send_list, recv_list)
            cols = t.all_to_all_v(ei[1][order], send_list, recv_list)
            w_local = None if w_part is None else t.all_to_all_v(w_part[order], send_list, recv_list)
        else:
            rows, cols, w_local = ei[0], ei[1], w_part
        if rows.numel() and (int(rows.min()) < lo or int(rows.max()) >= hi):
            raise L.TfgxError("edge routed to a rank that does not own its destination")
        local = torch.stack([rows - lo, cols]).to(torch.int32)
        assert int(local.shape[1]) == int((rp[hi] - rp[lo]).item())
        self._finish_build(local, w_local, torch.arange(int(local.shape[1]), device=local.device), rounds)
        return self

    def _finish_build(self, local, w_local, edge_ids, rounds):
        """Steps 3b-6 of the shard build from this rank's edges (`local`: [2, E_own], destinations already relative to
        own_lo, in this shard's edge order; `edge_ids`: what self.perm should point at)."""
        be = self.backend
        lo, hi = self.own_lo, self.own_hi
        self.num_edges = int(local.shape[1])
        self.row_ptr, col_slice, perm_local = be.build_csr(be.i32(local), self.n_own, self.n_global)
        w_slice = None if w_local is None else be.permute_rows(w_local, perm_local)
        ids_dev = be.i32(edge_ids)
        self.perm = ids_dev[perm_local.long()].contiguous()     # CSR position -> global edge id (caller's order)

        # 4. halo: remote sources, sorted + de-duplicated; col remapped into [own | halo].  A peer whose block is
        #    referenced to >= TFGX_DENSE_PEER_PCT percent (default 90; uniform random graphs at 8 GPUs: 99.8) is requested
        #    WHOLE: its owner then sends the block as it is — no pack kernel, no send buffer.
        self_halo = self._self_halo_rows is not None
        src_hi = lo + min(max(self._self_halo_rows, 0), self.n_own) if self_halo else hi
        dense_pct = int(os.environ.get("TFGX_DENSE_PEER_PCT", "90"))
        self.halo_ids, col_local = be.halo_plan(col_slice, lo, src_hi, self.n_global, self.n_own,
                                                peer_bounds=self.bounds, rank=self.rank, dense_pct=dense_pct)
        self.n_halo = int(self.halo_ids.shape[0])
        self.n_table = self.n_own + self.n_halo

        # 5. who sends what: per-peer request counts -> id lists, then cut every peer's list into `rounds` slices:
        #    the halo travels in R all-to-all-v rounds and the edges of round j are reduced while round j+1 flies
        if rounds is None:
            rounds = int(os.environ.get("TFGX_HALO_ROUNDS", "4" if self.n_global // max(self.world, 1) >= 16384 else "1"))
        self.rounds = max(1, min(int(rounds), 16)) if (self.world > 1 or self_halo) else 0
        self._build_exchange_lists()
        col_local = self._round_major_layout(col_local)

        # 6. per-row stable partition [own-source edges | round-0 halo edges | round-1 halo edges | ...]
        self.n_class = self.rounds + 1
        class_bounds = [self.n_own + int(v) for v in self.round_offset[1:self.rounds]] if self.rounds > 1 else []
        self.rpk, self.col, self.w = be.split_by_class(self.row_ptr, col_local, w_slice, [self.n_own] + class_bounds
                                                       if self.rounds >= 1 else [], self.n_class)
        self.in_degree = (self.row_ptr[1:] - self.row_ptr[:-1]).contiguous()
        # long spans of any pass are reduced chunk-wise (skewed graphs); None on near-regular graphs
        hub_fn = getattr(be, "hub_lists", None)
        K1 = self.n_class
        self.hub = [hub_fn(self.rpk[k:], self.rpk[k + 1:], K1, self.n_own, self.num_edges) if hub_fn else None
                    for k in range(K1)]
        self.norm_w = None
        self.self_coef = None

    def _build_exchange_lists(self):
        """Per-peer request counts and id lists, on the device (the halo ids are node-sized: 10^8 at papers100M shape).
        A peer block requested whole is DENSE: its ids are not exchanged at all (the owner knows them: its own range)."""
        be, W, t = self.backend, self.world, self.transport
        dev = self.halo_ids.device
        cuts = torch.searchsorted(self.halo_ids.long(), torch.as_tensor(self.bounds, dtype=torch.int64, device=dev)).tolist()
        self.recv_counts = [int(cuts[p + 1] - cuts[p]) for p in range(W)]       # rows I receive from p
        assert self.recv_counts[self.rank] == 0 or self._self_halo_rows is not None
        size = [int(self.bounds[p + 1] - self.bounds[p]) for p in range(W)]
        dense_recv = [self.recv_counts[p] == size[p] and size[p] > 0 and p != self.rank for p in range(W)]
        if self.rounds == 0:
            self.send_counts, self.dense_send = [0] * W, [False] * W
            self._give_local, self._give_start = be.i32(np.zeros(0, np.int32)), [0] * (W + 1)
            return
        rc = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
        self.send_counts = [int(v) for v in t.all_to_all_v(rc, [1] * W, [1] * W).tolist()]   # rows p wants from me
        # a request for exactly n_own rows can only be the whole block [own_lo, own_hi): no list needed, no pack
        self.dense_send = [self.send_counts[q] == self.n_own and self.n_own > 0 and q != self.rank for q in range(W)]
        want_counts = [0 if dense_recv[p] else self.recv_counts[p] for p in range(W)]
        give_counts = [0 if self.dense_send[q] else self.send_counts[q] for q in range(W)]
        seg = [self.halo_ids[cuts[p]:cuts[p + 1]] for p in range(W) if not dense_recv[p]]
        want = torch.cat(seg) if seg else self.halo_ids[:0]
        give = t.all_to_all_v(want.contiguous(), want_counts, give_counts)       # global ids peers want from me
        if give.numel() and (int(give.min()) < self.own_lo or int(give.max()) >= self.own_hi):
            raise L.TfgxError("a peer asked for rows this rank does not own")
        self._give_local = (give - self.own_lo).to(torch.int32).contiguous()
        self._give_start = [0] + [int(v) for v in np.cumsum(give_counts)]

    def _round_major_layout(self, col_local):
        """Cut each peer's halo segment (and the matching send list) into `rounds` contiguous slices and lay the halo
        table out round-major: [round 0: peer 0 slice, peer 1 slice, ... | round 1: ...].  Sender and receiver use the
        same floor(L*j/R) cut points, so slice j of the list peer q requested is slice j of what q expects.  Device ops
        only (W x R small host loops that build index ranges)."""
        be, W, R = self.backend, self.world, self.rounds
        dev = self.halo_ids.device
        if R == 0:
            self.round_offset = np.zeros(1, dtype=np.int64)
            self.round_recv_counts, self.round_send_counts, self.round_send_dense, self.round_send_idx = [], [], [], []
            self.send_idx_packed = be.i32(np.zeros(0, np.int32))
            return col_local
        cut = lambda length, j: (length * j) // R                                  # noqa: E731
        seg_start = np.concatenate([[0], np.cumsum(self.recv_counts)]).astype(np.int64)   # old halo layout: by peer
        self.round_recv_counts = [[cut(self.recv_counts[p], j + 1) - cut(self.recv_counts[p], j) for p in range(W)]
                                  for j in range(R)]
        self.round_send_counts = [[cut(self.send_counts[p], j + 1) - cut(self.send_counts[p], j) for p in range(W)]
                                  for j in range(R)]
        # dense (round, peer) entries: the contiguous own rows [cut(n_own, j), cut(n_own, j + 1)); -1 = packed
        self.round_send_dense = [[cut(self.send_counts[p], j) if self.dense_send[p] else -1 for p in range(W)]
                                 for j in range(R)]
        self.round_offset = np.concatenate([[0], np.cumsum([sum(c) for c in self.round_recv_counts])]).astype(np.int64)
        pieces = [torch.arange(int(seg_start[p]) + cut(self.recv_counts[p], j), int(seg_start[p]) + cut(self.recv_counts[p], j + 1),
                               device=dev) for j in range(R) for p in range(W)]
        order = torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.int64, device=dev)   # table order -> old position
        newpos = torch.empty(self.n_halo, dtype=torch.int32, device=dev)
        newpos[order] = torch.arange(self.n_halo, dtype=torch.int32, device=dev)
        self.halo_ids = self.halo_ids[order].contiguous()                          # halo ids in table order
        self.round_send_idx = []
        for j in range(R):
            parts = [self._give_local[self._give_start[p] + cut(self.send_counts[p], j):
                                      self._give_start[p] + cut(self.send_counts[p], j + 1)]
                     for p in range(W) if not self.dense_send[p]]
            self.round_send_idx.append(torch.cat(parts).contiguous() if parts else self._give_local[:0])
        self.send_idx_packed = torch.cat(self.round_send_idx).contiguous() if self.round_send_idx else self._give_local[:0]
        # remap halo columns old position -> round-major position
        cl = col_local.long()
        is_halo = cl >= self.n_own
        remapped = torch.where(is_halo, self.n_own + newpos[(cl - self.n_own).clamp(min=0)].long(), cl)
        return remapped.to(torch.int32).contiguous()

    # ------------------------------------------------------------------ source table + exchange
    def alloc_table(self, num_features):
        """[n_own + n_halo, F] source table; rows [0, n_own) are this rank's own feature rows."""
        return self.backend.empty((self.n_table, int(num_features)))

    def own_rows(self, table):
        return table[:self.n_own]

    def halo_rows(self, table):
        return table[self.n_own:]

    def exchange_start(self, table):
        """Start the R rounds of the halo exchange into table[n_own:] (round-major) and return a handle for
        exchange_finish().  Product path (transport "tfgx_dist"): per round one pack launch for the packed peers on the
        compute stream, then grouped ncclSend / ncclRecv on the transport's communication stream — asynchronous, so the
        passes launched on the current stream afterwards overlap the transfer, round 0 completing first."""
        handles = self.transport.exchange_start(self, table)
        if isinstance(table, torch.Tensor) and not table.requires_grad:
            torch.autograd.graph.increment_version(table)      # halo rows arrive through raw pointers (RCCL / pack kernels)
        return handles

    def exchange_finish(self, handles, j=None):
        """Make the current stream wait for round j (None: all rounds)."""
        return self.transport.exchange_finish(self, handles, j)

    # ------------------------------------------------------------------ aggregation
    def aggregate(self, table, op=L.SUM, w="plan", self_coef=None, bias=None, act=L.ACT_NONE, out=None,
                  exchange=True, classes=None, split=None):
        """out[r] = reduce over ALL edges of own row r of w*table[col] (+ epilogue), halo exchange overlapped with
        the local-source pass.  `w`: "plan" = the shard's edge weights, None = unweighted, or a tensor in this
        shard's edge order.  `classes` (diagnostics, bench.py): run only these source classes (0 = own-source edges,
        j+1 = round-j halo edges) on whatever the table holds — used with exchange=False to time the passes alone."""
        be = self.backend
        w_t = self.w if (isinstance(w, str) and w == "plan") else w
        F = int(table.shape[1])
        if out is None:
            out = be.empty((self.n_own, F))
        # exchange: True = start it here; a list = handles of an exchange the caller already started; False = none
        handles = exchange if isinstance(exchange, list) else (self.exchange_start(table) if exchange else None)
        K1, rpk = self.n_class, self.rpk
        for k in (range(K1) if classes is None else classes):
            last = k == K1 - 1
            if k >= 1:
                self.exchange_finish(handles, k - 1)          # class k reads the rows of round k-1
            kw = {"hub": self.hub[k]} if self.hub[k] is not None else {}
            if split is not None:
                kw["split"] = split
            if last:      # epilogue (self-loop term, mean divisor, bias, activation) once, in the final pass
                be.segment_reduce(rpk[k:], rpk[k + 1:], K1, self.col, w_t, self.n_own, table, out, op, act=act,
                                  accumulate=k > 0, self_coef=self_coef, bias=bias,
                                  mean_count=self.in_degree if op == L.MEAN else None, **kw)
            else:         # class 0 (own-source edges) overlaps the whole exchange, class j+1 overlaps rounds > j
                be.segment_reduce(rpk[k:], rpk[k + 1:], K1, self.col, w_t, self.n_own, table, out,
                                  L.SUM if op == L.MEAN else op, accumulate=k > 0, **kw)
        return out

    # ------------------------------------------------------------------ static input features (layer 0)
    def prepare_static_features(self, x_own):
        """EXPLICIT opt-in, the sharded twin of tfg.prepare_static_features: `x_own` are this rank's rows
        of the dataset's input features, which never change.  The halo is exchanged ONCE, the [own | halo] table is kept,
        and — at widths where a row straddles an extra 128-byte line (F = 100) — so is its edge-resident-tail layout in
        this shard's CSR order.  Every later layer-0 aggregation (`aggregate_static`) runs without any exchange.
        Returns the handle; its `bytes` entry reports what it holds."""
        be = self.backend
        table = self.alloc_table(int(x_own.shape[1]))
        self.own_rows(table).copy_(x_own)
        self.exchange_finish(self.exchange_start(table))
        split_fn = getattr(be, "split_rows", None)
        split = split_fn(table, self.col) if split_fn is not None else None
        nbytes = 4 * int(table.numel()) + (0 if split is None else 4 * sum(int(t.numel()) for t in split))
        return {"table": table, "split": split, "bytes": nbytes}

    def aggregate_static(self, static, op=L.SUM, w="plan", self_coef=None, bias=None, act=L.ACT_NONE, out=None):
        """aggregate() over a table prepared by prepare_static_features: no exchange, static layout when present."""
        return self.aggregate(static["table"], op, w=w, self_coef=self_coef, bias=bias, act=act, out=out, exchange=False,
                              split=static["split"])

    # ------------------------------------------------------------------ feature-column-chunked halo
    def aggregate_chunked(self, x_own, num_splits, op=L.SUM, w="plan", self_coef=None, bias=None, act=L.ACT_NONE):
        """aggregate() with the feature columns cut into `num_splits` chunks (the reference's own remedy for graphs
        whose [E, F] / [N, F] intermediates do not fit: `num_splits`, layers/conv/gcn.py:21-23 ->
        utils/tf_sparse_utils.py:71-90 -> SparseMatrix.matmul's column splits, nn/conv/gcn.py:274-280 — "does not
        affect the output").  Here it bounds the HALO: only (n_own + n_halo) x chunk_width floats of source table exist
        at a time instead of (n_own + n_halo) x F — at papers100M shape the 49.7 GB halo of a shard becomes 49.7 /
        num_splits GB.  Two chunk tables are alive so that chunk c+1's exchange overlaps chunk c's passes."""
        be = self.backend
        F = int(x_own.shape[1])
        sizes = compute_num_or_size_splits(F, num_splits)
        if sizes is None:
            sizes = [F]
        elif isinstance(sizes, int):
            sizes = [F // sizes] * sizes
        out = be.empty((self.n_own, F))
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.last_chunk_table_floats = 0

        def launch(c):
            lo, hi = int(starts[c]), int(starts[c + 1])
            table = self.alloc_table(hi - lo)
            self.last_chunk_table_floats = max(self.last_chunk_table_floats, int(table.numel()))
            self.own_rows(table).copy_(x_own[:, lo:hi])
            return table, self.exchange_start(table)

        nxt = launch(0)
        for c in range(len(sizes)):
            table, handles = nxt
            nxt = launch(c + 1) if c + 1 < len(sizes) else None
            lo, hi = int(starts[c]), int(starts[c + 1])
            b = None if bias is None else bias[lo:hi].contiguous()
            self.aggregate(table, op, w=w, self_coef=self_coef, bias=b, act=act, out=out[:, lo:hi], exchange=handles)
        return out

    # ------------------------------------------------------------------ training (backward of the sharded aggregation)
    def _transposed_local(self):
        """CSR of this shard's edges by SOURCE-TABLE index (own rows, then halo rows): row t lists the own destination
        rows its table row feeds.  (row_ptr_t [n_table+1], dst_t [E], perm_t [E]: transposed position -> position in
        self.col / self.w order).  Built once, on the first backward."""
        if getattr(self, "_tl", None) is None:
            be = self.backend
            deg = self.in_degree.long()
            rows = torch.repeat_interleave(torch.arange(self.n_own, device=deg.device), deg).to(torch.int32)
            self._tl = be.build_csr(torch.stack([self.col, rows]), max(self.n_table, 1), max(self.n_own, 1))
        return self._tl

    def reverse_exchange(self, d_table, inplace=False, started=None):
        """Gradient w.r.t. the [own | halo] source table -> gradient w.r.t. this rank's own rows (the backward of the
        halo exchange).  The halo-row gradients travel back along the forward exchange's lists (what I received from p in
        round j, I send to p; what I sent, I receive), one grouped exchange per round; returned rows are added into the own
        rows at the forward send indices, peer by peer in rank order and round by round — a fixed order, and one peer's
        list has no repeated row, so the sum is deterministic without atomics (tfgx_scatter_add_rows_f32)."""
        d_table = d_table.contiguous()
        handle = started if started is not None else self.transport.reverse_start(self, d_table)
        d_own = d_table[:self.n_own]
        if not inplace:
            d_own = d_own.clone()
        return self.transport.reverse_finish(self, handle, d_own)

    def halo_table(self, h_own, defer=False):
        """Differentiable [own | halo] source table of own rows `h_own`: forward = the halo exchange, backward =
        reverse_exchange.  Any single-GPU differentiable operator on local_plan() composed with it is the sharded
        operator WITH its backward (max aggregation, the fused attention).  defer=True: the exchange is only STARTED
        (asynchronous on the communication stream); the caller launches whatever row-local work it has and calls
        halo_table_wait() before the first kernel that reads halo rows."""
        return _HaloGather.apply(self, h_own, bool(defer))

    def halo_table_wait(self):
        h = getattr(self, "_deferred_exchange", None)
        if h is not None:
            self._deferred_exchange = None
            with torch.no_grad():          # (the host-staged test transport copies into views made inside the Function)
                self.exchange_finish(h)

    def pop_deferred_exchange(self):
        """The handle of the exchange halo_table(defer=True) started (the caller now owns the waits), or None."""
        h = getattr(self, "_deferred_exchange", None)
        self._deferred_exchange = None
        return h

    def tracked_max_passes(self, table, handles):
        """The forward of a TRAINABLE max aggregation as one tracked launch per source class (plan.can_track conditions on
        the table; no chunked long spans), or None when the shard / width does not allow it.  -> callable(x2, w, out, packed)."""
        from ..plan import can_track
        x2, ldx = L.row_major_2d(table)
        if any(h is not None for h in self.hub) or not can_track(self.local_plan(), x2, ldx):
            return None
        be, K1, rpk = self.backend, self.n_class, self.rpk

        token = int(table.data_ptr())

        start_reverse = self._early_reverse_rounds(token, "max_halo_first_backwards", check_ptr=True)

        def run(x2_, w_, out, packed):
            self.counters["max_span_forwards"] = self.counters.get("max_span_forwards", 0) + 1
            with torch.no_grad():
                for k in range(K1):
                    if k >= 1:
                        self.exchange_finish(handles, k - 1)      # class k reads the rows of round k - 1
                    be.segment_reduce(rpk[k:], rpk[k + 1:], K1, self.col, w_, self.n_own, x2_, out, L.MAX,
                                      accumulate=k > 0, track=packed, track_row_begin=rpk)
        run.halo_first = (self.n_own, self.halo_round_windows(), start_reverse)
        return run

    def halo_round_windows(self):
        """Table-row windows [lo, hi) of the halo rows, one per exchange round (the halo part of the table is round-major)."""
        return [(self.n_own + int(self.round_offset[j]), self.n_own + int(self.round_offset[j + 1]))
                for j in range(self.rounds)]

    def _early_reverse_rounds(self, token, counter, check_ptr):
        """callable(j, d_table) for the backward of a trainable aggregation on the [own | halo] table: called once per round,
        in order, as soon as round j's rows of d_table are final — they start travelling (tfgx_halo_reverse_start_round); after
        the last round the _HaloGather node of THIS table (matched by its storage `token`) finds the exchange started and only
        finishes it."""
        state = {"handle": None}

        def start_round(j, d_table):
            state["handle"] = self.transport.reverse_start_round(self, d_table, j, state["handle"])
            if j == self.rounds - 1:
                self.counters[counter] = self.counters.get(counter, 0) + 1
                self._early_reverse = (token, state["handle"], d_table if check_ptr else None,
                                       None if check_ptr else d_table)
                state["handle"] = None
        return start_round

    def local_plan(self):
        """This shard as an [n_own x n_table] CSR operator (plan.CsrPlan over row_ptr / col — rows stay contiguous
        through the per-class partition), with the transposed plan available for the backward passes."""
        if getattr(self, "_local_plan", None) is None:
            from ..plan import CsrPlan
            rows = torch.repeat_interleave(torch.arange(self.n_own, device=self.col.device),
                                           self.in_degree.long()).to(torch.int32)
            self._local_plan = CsrPlan.from_sorted(self.row_ptr, self.col, max(self.n_table, 1),
                                                   edge_index=torch.stack([rows, self.col]))
        return self._local_plan

    def aggregate_backward(self, g_out, w="plan", self_coef=None, mean=False):
        """d(loss)/d(own table rows) of out = aggregate(table, SUM | MEAN, w, self_coef) given g_out = d(loss)/d(out).

        1. local transposed pass: dT[t] = sum over this shard's edges with source t of w * g[row] — for own rows AND for
           halo rows (gradients that belong to peers);
        2. reverse_exchange: halo-row gradients go back to their owners and are accumulated there in a fixed order;
        3. the implicit self-loop term self_coef[r] * g[r]."""
        be = self.backend
        w_t = self.w if (isinstance(w, str) and w == "plan") else w
        g = g_out.contiguous()
        if mean:
            g = g / self.in_degree.clamp(min=1).to(g.dtype).unsqueeze(1)
        U = int(g.shape[1])
        rp_t, dst_t, perm_t = self._transposed_local()
        wt = None if w_t is None else be.permute_rows(w_t, perm_t)
        d_table = be.empty((max(self.n_table, 1), U))
        n_own, n_halo = self.n_own, self.n_halo
        if getattr(self, "_tl_hub", None) is None:      # hub SOURCES of the shard: chunked, as the forward's hub rows are
            hub_fn = getattr(be, "hub_lists", None)
            self._tl_hub = [(hub_fn(rp_t[a:], rp_t[a + 1:], 1, m, self.num_edges) if (hub_fn and m > 0) else None) or False
                            for a, m in ((n_own, n_halo), (0, n_own))]
        # halo rows FIRST: their gradients belong to peers and start travelling (reverse exchange, asynchronous on the
        # communication stream) while the rest of the transposed pass runs on the compute stream — round by round (the halo
        # table is round-major): round j's window, then round j's sends, so each round is on the wire under the windows that
        # follow; hub sources in the halo (chunk lists span the whole halo range) keep one window and one start
        if n_halo > 0 and self._tl_hub[0]:
            be.segment_reduce(rp_t[n_own:], rp_t[n_own + 1:], 1, dst_t, wt, n_halo, g, d_table[n_own:n_own + n_halo], L.SUM,
                              hub=self._tl_hub[0])
            handle = self.transport.reverse_start(self, d_table)
        else:
            handle = None
            for j in range(self.rounds):
                lo, hi = n_own + int(self.round_offset[j]), n_own + int(self.round_offset[j + 1])
                if hi > lo:
                    be.segment_reduce(rp_t[lo:], rp_t[lo + 1:], 1, dst_t, wt, hi - lo, g, d_table[lo:hi], L.SUM)
                handle = self.transport.reverse_start_round(self, d_table, j, handle)
        if n_own > 0:
            kw = {"hub": self._tl_hub[1]} if self._tl_hub[1] else {}
            be.segment_reduce(rp_t, rp_t[1:], 1, dst_t, wt, n_own, g, d_table[:n_own], L.SUM, **kw)
        d_own = self.transport.reverse_finish(self, handle, d_table[:n_own])
        if self_coef is not None:
            d_own = d_own + self_coef.unsqueeze(1) * g
        return d_own

    def aggregate_trainable(self, h_own, op=L.SUM, w="plan", self_coef=None):
        """Differentiable sharded aggregation of own rows `h_own` [n_own, U] (torch autograd; forward = aggregate with the
        overlapped halo exchange, backward = aggregate_backward with the reverse exchange)."""
        if op in (L.SUM, L.MEAN):
            return _ShardedAggregate.apply(self, op, w, self_coef, h_own)
        if self_coef is not None:
            raise NotImplementedError("max aggregation with an implicit self-loop is inference-only")
        # max (the reducer of max_pool_graph_sage): differentiable halo table, then the single-GPU max aggregation with
        # its tie-sharing gradient on the shard's rectangular plan; halo-row gradients return through reverse_exchange
        w_t = self.w if (isinstance(w, str) and w == "plan") else w
        table = self.halo_table(h_own, defer=True)     # the rows start travelling; the own-source span runs under them
        return self.backend.aggregate_autograd(self, table, op, w_t, handles=self.pop_deferred_exchange())

    def gat_trainable(self, x_own, query_kernel, query_bias, query_act, key_kernel, key_bias, key_act, kernel, bias=None,
                      activation=None, num_heads=1):
        """Sharded GAT layer (nn/conv/gat.py:13-122, split_value_heads=True) whose output carries gradients to the five
        weights and to x_own: Q stays local, [K | V] rows travel once as ONE differentiable halo table, the fused
        attention runs on the shard's rectangular plan, and the d[K | V] of halo rows return to their owners in the
        backward (reverse_exchange).  query_act / key_act: L.ACT_* codes (as gat()); activation: a callable or None."""
        be = self.backend
        A = int(query_kernel.shape[1])
        K = be.linear(x_own, key_kernel, key_bias, key_act)
        V = be.linear(x_own, kernel)
        table = self.halo_table(torch.cat([K, V], 1), defer=True)      # [K | V] rows start travelling ...
        Q = be.linear(x_own, query_kernel, query_bias, query_act)      # ... while the row-local Q projection runs
        out = be.gat_attention_autograd(self, Q, table[:, :A], table[:, A:], num_heads,      # ... and the own-source span
                                        handles=self.pop_deferred_exchange())
        if bias is not None:
            out = _bias_add(out, bias)
        return activation(out) if activation is not None else out

    def pool_graph_sage_trainable(self, x_own, self_kernel, neighbor_mlp_kernel, neighbor_kernel, neighbor_mlp_bias=None,
                                  bias=None, act=L.ACT_NONE, concat=True, op=L.MAX):
        """pool_graph_sage with gradients (nn/conv/graph_sage.py:164-287): the per-node MLP rows are the halo table; the
        SAME activation after the MLP and at the end, as the reference applies it."""
        be = self.backend
        h = be.linear(x_own, neighbor_mlp_kernel, neighbor_mlp_bias, act)
        if op in (L.SUM, L.MEAN) and int(neighbor_kernel.shape[1]) < int(h.shape[1]):
            # linear reducer: project first, ku-wide rows travel (forward) and ku-wide gradients come back (backward)
            a = be.linear(x_own, self_kernel)
            b = self.aggregate_trainable(be.linear(h, neighbor_kernel), op, w=None)
            out = torch.cat([a, b], 1) if concat else a + b
            if bias is not None:
                out = _bias_add(out, bias)
            return torch.relu(out) if act == L.ACT_RELU else out
        if op in (L.SUM, L.MEAN):
            a = be.linear(x_own, self_kernel)
            reduced = self.aggregate_trainable(h, op, w=None)
        else:
            table = self.halo_table(h, defer=True)                     # the MLP rows start travelling ...
            a = be.linear(x_own, self_kernel)                          # ... under the row-local self projection
            reduced = be.aggregate_autograd(self, table, op, None,     # ... and under the own-source span of the max
                                            handles=self.pop_deferred_exchange())
        b = be.linear(reduced, neighbor_kernel)
        out = torch.cat([a, b], 1) if concat else a + b
        if bias is not None:
            out = _bias_add(out, bias)
        return torch.relu(out) if act == L.ACT_RELU else out

    def gcn_trainable(self, x_own, kernel, bias=None, activation=None):
        """Sharded GCN layer whose output carries gradients to kernel / bias / x_own (nn/conv/gcn.py:225-290 under a
        tf.GradientTape in the reference's training loops).  Weights are replicated: call all_reduce_gradients() after
        backward() — the one collective the reference's distributed demos perform (demo_distributed_gcn.py:52-57)."""
        if self.norm_w is None:
            self.build_gcn_norm()
        h = x_own if kernel is None else self.backend.linear(x_own, kernel)
        out = self.aggregate_trainable(h, L.SUM, w=self.norm_w, self_coef=self.self_coef)
        if bias is not None:
            out = _bias_add(out, bias)
        return activation(out) if activation is not None else out

    def all_reduce_gradients(self, params):
        """Sum the gradients of replicated weights over the ranks, bucketed into ONE flat all-reduce (the weights of
        this path are KB-sized; a collective per tensor would be latency-bound on xGMI)."""
        grads = [p.grad for p in params if p.grad is not None]
        if self.world == 1 or not grads:
            return
        flat = self.transport.all_reduce_sum_f32(torch.cat([g_.reshape(-1) for g_ in grads]).contiguous())
        off = 0
        for g_ in grads:
            g_.copy_(flat[off:off + g_.numel()].reshape(g_.shape))
            off += g_.numel()

    # ------------------------------------------------------------------ GCN
    def build_gcn_norm(self, norm="both", add_self_loop=True, sym=True, renorm=True, improved=False):
        """Sharded gcn_norm_adj (nn/conv/gcn.py:32-130): row degrees are local to the owner; the degrees of halo sources
        arrive through one 1-column halo exchange.  sym=False (:87-91: the right factor uses COLUMN sums): a column sum
        collects weights of edges that live on other ranks, i.e. it is exactly the backward of a 1-column aggregation
        of ones — local transposed pass, reverse halo exchange, owner-side accumulate (aggregate_backward)."""
        be = self.backend
        fill = 2.0 if improved else 1.0
        if norm == "both":
            diag = fill if (add_self_loop and renorm) else 0.0
        else:
            diag = fill if add_self_loop else 0.0
        deg_table = self.alloc_table(1)
        # degree over ALL edges of the row: local + halo parts are contiguous per row in self.w / self.col
        own_deg = be.weight_sum(self.row_ptr, self.w, self.n_own, diag)
        self.own_rows(deg_table)[:, 0] = own_deg
        self.exchange_finish(self.exchange_start(deg_table))
        mode = L.NORM_MODES[norm]
        col_table = None
        if norm == "both" and not sym:
            ones = be.f32(np.ones((self.n_own, 1), np.float32))
            col_own = self.aggregate_backward(ones, w="plan" if self.w is not None else None)[:, 0] + diag
            col_table = self.alloc_table(1)
            self.own_rows(col_table)[:, 0] = col_own
            self.exchange_finish(self.exchange_start(col_table))
            col_table = col_table[:, 0].contiguous()
        self.norm_w, sc = be.gcn_norm_edges(self.row_ptr, self.col, self.w, self.n_own,
                                            deg_table[:, 0].contiguous(), mode, fill, add_self_loop, renorm,
                                            col_deg=col_table)
        self.self_coef = sc if add_self_loop else None
        return self

    def gcn_propagate(self, table, bias=None, act=L.ACT_NONE, out=None):
        """A_hat @ h for own rows (h = table's own rows; halo rows are fetched here)."""
        if self.norm_w is None:
            self.build_gcn_norm()
        return self.aggregate(table, L.SUM, w=self.norm_w, self_coef=self.self_coef, bias=bias, act=act, out=out)

    def gcn(self, x_own, kernel, bias=None, act=L.ACT_NONE):
        """Sharded GCN layer (nn/conv/gcn.py:225-290): the GEMM is row-local and writes straight into the source
        table, so only `units`-wide rows travel."""
        be = self.backend
        if kernel is None:
            table = self.alloc_table(int(x_own.shape[1]))
            self.own_rows(table).copy_(x_own)
        else:
            table = self.alloc_table(int(kernel.shape[1]))
            be.gemm_bias_act(x_own, kernel, out=self.own_rows(table))
        return self.gcn_propagate(table, bias=bias, act=act)

    # ------------------------------------------------------------------ GAT
    def gat(self, x_own, query_kernel, query_bias, query_act, key_kernel, key_bias, key_act, kernel, bias=None,
            act=L.ACT_NONE, num_heads=1):
        """Sharded GAT layer (nn/conv/gat.py:13-122, split_value_heads=True): Q stays local; K and V of the halo
        sources travel together in one [A + U]-wide exchange; the local-source edges are reduced to a raw
        online-softmax state while the exchange is in flight, the halo-source edges afterwards, and the two states
        plus the appended self-loop edge are merged (tfgx_gat_merge_passes_f32)."""
        be = self.backend
        A, U = int(query_kernel.shape[1]), int(kernel.shape[1])
        Q = be.gemm_bias_act(x_own, query_kernel, bias=query_bias, act=query_act)
        table = self.alloc_table(A + U)
        be.gemm_bias_act(x_own, key_kernel, bias=key_bias, act=key_act, out=self.own_rows(table)[:, :A])
        be.gemm_bias_act(x_own, kernel, out=self.own_rows(table)[:, A:])
        handle = self.exchange_start(table)
        return self._gat_attention_spans(Q, table[:, :A], table[:, A:], num_heads, handle, bias, act, None)

    def _gat_attention_spans(self, Q, K, V, num_heads, handle, bias, act, out, stats=None):
        """The attention over this shard's rows as span passes + merge, `handle` = the exchange of the [K | V] table in
        flight (the own-source span runs under it).  stats: [n_own, 2 H] receives the merged softmax statistics."""
        be = self.backend
        U = int(V.shape[1])
        K1, rpk = self.n_class, self.rpk
        if stats is not None:
            self.counters["gat_span_training_forwards"] = self.counters.get("gat_span_training_forwards", 0) + 1
        spans = [(rpk, rpk[1:])] + ([(rpk[1:], rpk[K1:])] if K1 > 1 else [])   # own-source span | all halo classes
        plan = self._gat_parts(spans)
        kw = {} if stats is None else {"stats": stats}
        if out is None:
            out = be.empty((self.n_own, U))
        if plan is None:        # near-regular shard: two raw states per row, fixed-stride merge
            s_acc = be.empty((2 * self.n_own, U))
            s_ml = be.empty((2 * self.n_own, 2 * num_heads))
            be.gat_pass(rpk, rpk[1:], K1, self.col, self.n_own, Q, K, V, num_heads, s_acc[:self.n_own], s_ml[:self.n_own])
            self.exchange_finish(handle)
            n_passes = 1
            if K1 > 1:      # all halo classes are one contiguous span per row: [rpk[r*K1+1], rpk[(r+1)*K1])
                be.gat_pass(rpk[1:], rpk[K1:], K1, self.col, self.n_own, Q, K, V, num_heads, s_acc[self.n_own:],
                            s_ml[self.n_own:])
                n_passes = 2
            return be.gat_merge(Q, K, V, num_heads, self.n_own, s_acc, s_ml, n_passes, bias, act, out, **kw)
        # skewed shard: long spans of either pass are cut into chunks (as single-GPU hub rows are); every row then merges
        # a LIST of parts — whole passes and chunks — named by (part_ptr, part_idx)
        n_states = plan["n_states"]
        s_acc = be.empty((n_states, U))
        s_ml = be.empty((n_states, 2 * num_heads))
        for t, (rb, re) in enumerate(spans):
            if t == 1:
                self.exchange_finish(handle)
            lo = t * self.n_own
            hub = plan["hub"][t]
            be.gat_pass(rb, re, K1, self.col, self.n_own, Q, K, V, num_heads, s_acc[lo:lo + self.n_own],
                        s_ml[lo:lo + self.n_own], skip_longer_than=hub[0] if hub is not None else 0)
            if hub is not None:
                thr, hub_rows, chunk_ptr, chunk_begin, chunk_end, chunk_row = hub
                c0, nc = plan["chunk_off"][t], int(chunk_begin.shape[0])
                be.gat_pass(chunk_begin, chunk_end, 1, self.col, nc, Q, K, V, num_heads, s_acc[c0:c0 + nc],
                            s_ml[c0:c0 + nc], part_row=chunk_row)
        if len(spans) == 1:
            self.exchange_finish(handle)
        return be.gat_merge_parts(Q, K, V, num_heads, self.n_own, s_acc, s_ml, plan["part_ptr"], plan["part_idx"], bias,
                                  act, out, **kw)

    def _gat_parts(self, spans):
        """Part lists for the sharded GAT on a skewed shard, or None when no span of either pass is long.  State rows:
        [pass 0 rows | pass 1 rows | pass 0 chunks | pass 1 chunks]; row r merges, per pass, either its whole-pass state
        or the states of its chunks."""
        if getattr(self, "_gat_parts_cache", None) is not None:
            return self._gat_parts_cache or None
        be = self.backend
        hub_fn = getattr(be, "hub_lists", None)
        hubs = [hub_fn(rb, re, self.n_class, self.n_own, self.num_edges) if hub_fn else None for rb, re in spans]
        if all(h is None for h in hubs):
            self._gat_parts_cache = False
            return None
        dev = self.row_ptr.device
        n, T = self.n_own, len(spans)
        cnt = torch.ones((T, n), dtype=torch.int64, device=dev)
        chunk_off, off = [], T * n
        for t, h in enumerate(hubs):
            chunk_off.append(off)
            if h is not None:
                _, hub_rows, chunk_ptr, chunk_begin, _, _ = h
                cnt[t, hub_rows.long()] = (chunk_ptr[1:] - chunk_ptr[:-1]).long()
                off += int(chunk_begin.shape[0])
        per_row = cnt.sum(0)
        part_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        part_ptr[1:] = torch.cumsum(per_row, 0)
        part_idx = torch.empty(int(part_ptr[-1].item()), dtype=torch.int64, device=dev)
        start = part_ptr[:-1].clone()
        for t, h in enumerate(hubs):
            whole = torch.ones(n, dtype=torch.bool, device=dev)
            if h is not None:
                _, hub_rows, chunk_ptr, _, _, _ = h
                hr = hub_rows.long()
                whole[hr] = False
                nch = (chunk_ptr[1:] - chunk_ptr[:-1]).long()
                owner = torch.repeat_interleave(torch.arange(hr.shape[0], device=dev), nch)
                k = torch.arange(int(nch.sum().item()), device=dev) - chunk_ptr[:-1].long()[owner]
                part_idx[start[hr][owner] + k] = chunk_off[t] + chunk_ptr[:-1].long()[owner] + k
            rows = torch.nonzero(whole).flatten()
            part_idx[start[rows]] = t * n + rows
            start = start + cnt[t]
        self._gat_parts_cache = dict(hub=hubs, chunk_off=chunk_off, n_states=off, part_ptr=part_ptr.to(torch.int32).contiguous(),
                                     part_idx=part_idx.to(torch.int32).contiguous())
        return self._gat_parts_cache

    # ------------------------------------------------------------------ GraphSAGE reduce
    def neighbor_reduce(self, x_own, op, weighted=True):
        """mean / sum / max of (w *) x[col] over in-edges of own rows (nn/conv/graph_sage.py:34-41)."""
        table = self.alloc_table(int(x_own.shape[1]))
        self.own_rows(table).copy_(x_own)
        return self.aggregate(table, op, w="plan" if (weighted and self.w is not None) else None)

    # ------------------------------------------------------------------ GraphSAGE layers
    def _sage_combine(self, x_own, reduced, self_kernel, neighbor_kernel, bias, act, concat):
        """x @ self_kernel (concat | +) reduced @ neighbor_kernel, + bias, activation (graph_sage.py:43-58); both
        GEMMs are row-local.  With concat they write into the two halves of the output."""
        be = self.backend
        ku_x, ku_n = int(self_kernel.shape[1]), int(neighbor_kernel.shape[1])
        if concat:
            h = be.empty((self.n_own, ku_x + ku_n))
            be.gemm_bias_act(x_own, self_kernel, bias=None if bias is None else bias[:ku_x], act=act, out=h[:, :ku_x])
            be.gemm_bias_act(reduced, neighbor_kernel, bias=None if bias is None else bias[ku_x:], act=act,
                             out=h[:, ku_x:])
            return h
        h = be.gemm_bias_act(x_own, self_kernel) + be.gemm_bias_act(reduced, neighbor_kernel)
        if bias is not None:
            h = h + bias
        return torch.relu(h) if act == L.ACT_RELU else h

    def graph_sage(self, x_own, self_kernel, neighbor_kernel, bias=None, act=L.ACT_NONE, concat=True, op=L.MEAN):
        """Sharded mean_graph_sage / sum_graph_sage (nn/conv/graph_sage.py:9-115): the weighted mean / sum over in-edges
        overlaps the exchange, the GEMMs are local.  mean and sum are linear, so when the neighbour projection is
        narrower than the input (units/2 < F: every hidden layer) it runs FIRST and only ku-wide rows travel and are
        gathered — the halo shrinks by F / ku; otherwise raw x rows travel once."""
        be = self.backend
        F, ku_x, ku_n = int(x_own.shape[1]), int(self_kernel.shape[1]), int(neighbor_kernel.shape[1])
        if not ku_n < F:
            reduced = self.neighbor_reduce(x_own, op, weighted=True)
            return self._sage_combine(x_own, reduced, self_kernel, neighbor_kernel, bias, act, concat)
        table = self.alloc_table(ku_n)
        be.gemm_bias_act(x_own, neighbor_kernel, out=self.own_rows(table))
        w = "plan" if self.w is not None else None
        if concat:
            h = be.empty((self.n_own, ku_x + ku_n))
            be.gemm_bias_act(x_own, self_kernel, bias=None if bias is None else bias[:ku_x], act=act, out=h[:, :ku_x])
            nb = self.aggregate(table, op, w=w, bias=None if bias is None else bias[ku_x:].contiguous(), act=act)
            h[:, ku_x:] = nb
            return h
        h = be.gemm_bias_act(x_own, self_kernel) + self.aggregate(table, op, w=w)
        if bias is not None:
            h = h + bias
        return torch.relu(h) if act == L.ACT_RELU else h

    def pool_graph_sage(self, x_own, self_kernel, neighbor_mlp_kernel, neighbor_kernel, neighbor_mlp_bias=None,
                        bias=None, act=L.ACT_NONE, concat=True, op=L.MAX):
        """Sharded mean_pool / max_pool_graph_sage (:164-287).  The reference overwrites the edge weights with ones, so
        the per-edge MLP act(x[col] @ W + b) is a per-NODE GEMM (as on one GPU): it runs on the owner and its output
        rows are what the halo exchange carries."""
        be = self.backend
        ku_x, ku_n, width = int(self_kernel.shape[1]), int(neighbor_kernel.shape[1]), int(neighbor_mlp_kernel.shape[1])
        if op == L.MEAN and ku_n < width:
            # the mean is linear: mean_j(h_j) @ W_neigh == mean_j(h_j @ W_neigh) (graph_sage.py:206-208) — both GEMMs run on
            # the owner and only ku-wide rows travel and are gathered (a quarter of the MLP width)
            hmlp = be.gemm_bias_act(x_own, neighbor_mlp_kernel, bias=neighbor_mlp_bias, act=act)
            table = self.alloc_table(ku_n)
            be.gemm_bias_act(hmlp, neighbor_kernel, out=self.own_rows(table))
            if concat:
                h = be.empty((self.n_own, ku_x + ku_n))
                be.gemm_bias_act(x_own, self_kernel, bias=None if bias is None else bias[:ku_x], act=act, out=h[:, :ku_x])
                h[:, ku_x:] = self.aggregate(table, L.MEAN, w=None, bias=None if bias is None else bias[ku_x:].contiguous(),
                                             act=act)
                return h
            h = be.gemm_bias_act(x_own, self_kernel) + self.aggregate(table, L.MEAN, w=None)
            if bias is not None:
                h = h + bias
            return torch.relu(h) if act == L.ACT_RELU else h
        table = self.alloc_table(width)
        be.gemm_bias_act(x_own, neighbor_mlp_kernel, bias=neighbor_mlp_bias, act=act, out=self.own_rows(table))
        reduced = self.aggregate(table, op, w=None)
        return self._sage_combine(x_own, reduced, self_kernel, neighbor_kernel, bias, act, concat)


def compute_num_or_size_splits(num_h_features, num_splits):
    """Same contract as the reference's helper (utils/tf_sparse_utils.py:71-90): None for no split, an int when the
    width divides evenly, else a list [ceil] * k + [remainder] that must have exactly num_splits entries."""
    if num_splits is None or num_splits == 1:
        return None
    if num_h_features % num_splits == 0:
        return int(num_splits)
    split_size = int(np.ceil(num_h_features / num_splits))
    num_pre = int(np.floor(num_h_features / split_size))
    last = num_h_features % split_size
    sizes = [split_size] * num_pre + ([last] if last > 0 else [])
    if len(sizes) != num_splits:
        raise Exception("cannot split H of shape [None, {}] into {} matrices, please provide a valid num_splits".format(
            num_h_features, num_splits))
    return sizes


class _ShardedAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sg, op, w, self_coef, h_own):
        table = sg.alloc_table(int(h_own.shape[1]))
        sg.own_rows(table).copy_(h_own.detach())
        ctx.sg, ctx.op, ctx.w, ctx.self_coef = sg, op, w, self_coef
        return sg.aggregate(table, op, w=w, self_coef=self_coef)

    @staticmethod
    def backward(ctx, g):
        d_own = ctx.sg.aggregate_backward(g, w=ctx.w, self_coef=ctx.self_coef, mean=ctx.op == L.MEAN)
        return None, None, None, None, d_own


class _HaloGather(torch.autograd.Function):
    """own rows -> [own | halo] table (forward: halo exchange; backward: ShardedGraph.reverse_exchange)."""

    @staticmethod
    def forward(ctx, sg, h_own, defer=False):
        ctx.sg = sg
        table = sg.alloc_table(int(h_own.shape[1]))
        ctx.table_ptr = int(table.data_ptr())
        early = getattr(sg, "_early_reverse", None)
        if early is not None and early[0] == ctx.table_ptr:
            sg._early_reverse = None       # a stale entry of an earlier table at this address (its backward was abandoned)
        sg.own_rows(table).copy_(h_own.detach())
        handle = sg.exchange_start(table)
        if defer:
            sg.halo_table_wait()                       # at most one deferred exchange per shard
            sg._deferred_exchange = handle
        else:
            sg.exchange_finish(handle)
        return table

    @staticmethod
    def backward(ctx, g_table):
        sg = ctx.sg
        early = getattr(sg, "_early_reverse", None)
        if early is not None and early[0] == ctx.table_ptr:
            # the consumer of this table (trainable max) already sent the halo rows of THIS gradient on their way
            sg._early_reverse = None
            if early[2] is not None and g_table.data_ptr() != early[2].data_ptr():
                raise RuntimeError("sharded max backward: the table gradient was replaced after its halo rows started "
                                   "travelling (the table has more than one consumer?)")
            if early[3] is not None and _DEBUG_CHECKS:
                # GAT halo-first backward (autograd assembles this gradient from the [K | V] slices, so the pointer differs):
                # the halo rows that already travelled must BE the halo rows of the final gradient — they are not if the
                # table had a second consumer.  A device comparison + host read: debug mode only (TFGX_DIST_DEBUG_CHECKS=1;
                # the multi-process tests run with it)
                if not torch.equal(g_table[sg.n_own:], early[3][sg.n_own:]):
                    raise RuntimeError("sharded GAT backward: the halo rows sent early differ from the table gradient's "
                                       "(the [K | V] table has more than one consumer?)")
            return None, sg.reverse_exchange(g_table, started=early[1]), None
        return None, sg.reverse_exchange(g_table.contiguous()), None
