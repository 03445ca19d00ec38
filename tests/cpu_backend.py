# coding=utf-8
"""numpy stand-in for tf_geometric_amd.dist.sharded.HipBackend — TEST INFRASTRUCTURE.

Lets the world_size-2 gloo tests exercise the sharding / halo-exchange ORCHESTRATION (partitioning, exchange
lists, two-pass accumulate, sharded GCN normalisation) on CPU.  Each method restates the contract of the C-ABI
entry point its HIP twin calls (include/tfgx.h); arithmetic is float64-accumulated like oracle/tfg_oracle.py.
"""
import numpy as np
import torch

SUM, MEAN, MAX = 0, 1, 2
FLT_LOWEST = np.float32(-3.4028234663852886e38)


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


class NumpyBackend(object):
    name = "numpy-test"
    device = torch.device("cpu")

    def i32(self, a):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        return torch.from_numpy(np.ascontiguousarray(a.astype(np.int32)))

    def f32(self, a):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))

    def empty(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype)

    def build_csr(self, edge_index, n_dst, n_src):
        ei = _np(edge_index).reshape(2, -1)
        assert ei.size == 0 or (ei[0].min() >= 0 and ei[0].max() < n_dst and ei[1].min() >= 0 and ei[1].max() < n_src)
        perm = np.argsort(ei[0], kind="stable").astype(np.int32)
        row_ptr = np.zeros(n_dst + 1, dtype=np.int32)
        np.cumsum(np.bincount(ei[0], minlength=n_dst), out=row_ptr[1:])
        return torch.from_numpy(row_ptr), torch.from_numpy(ei[1][perm].astype(np.int32)), torch.from_numpy(perm)

    def permute_rows(self, attr, perm):
        return self.f32(_np(self.f32(attr))[_np(perm)])

    def halo_plan(self, col, src_lo, src_hi, n_global, n_own, peer_bounds=None, rank=0, dense_pct=0):
        """tfgx_halo_mark / (dense-peer fill) / tfgx_halo_compact / tfgx_halo_remap_cols restated."""
        from tf_geometric_amd.dist.sharded import _fill_dense_peers
        c = _np(col)
        remote = (c < src_lo) | (c >= src_hi)
        flags = np.zeros(max(n_global, 1), np.int32)
        flags[c[remote]] = 1
        if peer_bounds is not None and dense_pct > 0 and n_global > 0:
            ft = torch.from_numpy(flags[:n_global])
            _fill_dense_peers(ft, peer_bounds, rank, dense_pct)
        ids = np.flatnonzero(flags[:n_global]).astype(np.int32)
        col_local = np.where(remote, n_own + np.searchsorted(ids, c), c - src_lo).astype(np.int32)
        return torch.from_numpy(ids), torch.from_numpy(col_local)

    def split_by_class(self, row_ptr, col_local, w, class_bounds, n_class):
        rp, c, wv = _np(row_ptr), _np(col_local), _np(w)
        n = rp.shape[0] - 1
        bounds = np.asarray(list(class_bounds), dtype=np.int64)
        rpk = np.zeros(n_class * n + 1, dtype=np.int32)
        c2 = np.empty_like(c)
        w2 = None if wv is None else np.empty_like(wv)
        for r in range(n):
            s, e = rp[r], rp[r + 1]
            cls = np.searchsorted(bounds, c[s:e], side="right") if n_class > 1 else np.zeros(e - s, np.int64)
            order = np.argsort(cls, kind="stable") + s
            c2[s:e] = c[order]
            if w2 is not None:
                w2[s:e] = wv[order]
            cnt = np.bincount(cls, minlength=n_class)
            rpk[r * n_class:(r + 1) * n_class] = s + np.concatenate([[0], np.cumsum(cnt)[:-1]])
        rpk[n * n_class] = rp[n]
        return torch.from_numpy(rpk), torch.from_numpy(c2), None if w2 is None else torch.from_numpy(w2)

    def gather_rows(self, x, idx, out=None):
        res = x[idx.long()]
        if out is not None:
            out.copy_(res)
            return out
        return res.contiguous()

    def scatter_add_rows(self, dst, idx, src):
        i = _np(idx).astype(np.int64)
        assert np.unique(i).size == i.size, "scatter_add_rows needs unique ids per call"
        d = _np(dst)
        d[i] += _np(src)
        return dst

    def linear(self, x, kernel, bias=None, act=0):
        h = x @ kernel                        # torch CPU autograd (test backend)
        h = h if bias is None else h + bias
        return torch.relu(h) if act == 1 else h

    @staticmethod
    def _rows_of(sg):
        deg = (sg.row_ptr[1:] - sg.row_ptr[:-1]).long()
        return torch.repeat_interleave(torch.arange(sg.n_own), deg)

    def aggregate_autograd(self, sg, table, op, w, handles=None):
        """max over the shard's edges in plain torch (scatter_reduce amax: tied maxima share the gradient evenly, as
        tf.math.unsorted_segment_max's gradient does)."""
        assert op == 2
        if handles is not None:          # (the HIP backend runs the own-source span under the exchange; here: wait first)
            with torch.no_grad():
                sg.exchange_finish(handles)
        rows, col = self._rows_of(sg), sg.col.long()
        msg = table[col] if w is None else table[col] * w.unsqueeze(1)
        out = torch.full((sg.n_own, table.shape[1]), FLT_LOWEST, dtype=table.dtype)
        return out.scatter_reduce(0, rows.unsqueeze(1).expand_as(msg), msg, reduce="amax", include_self=True)

    def gat_attention_autograd(self, sg, Q, K, V, num_heads, handles=None):
        """nn/conv/gat.py:40-122 over the shard's edges + the appended self-loop (source r = table row r), plain torch."""
        if handles is not None:
            with torch.no_grad():
                sg.exchange_finish(handles)
        n, H = sg.n_own, num_heads
        rows = torch.cat([self._rows_of(sg), torch.arange(n)])
        col = torch.cat([sg.col.long(), torch.arange(n)])
        d, dv = Q.shape[1] // H, V.shape[1] // H
        q, k, v = Q.view(n, H, d)[rows], K.reshape(-1, H, d)[col], V.reshape(-1, H, dv)[col]
        sc = (q * k).sum(-1) / float(np.sqrt(d))                                   # [E', H]
        mx = torch.full((n, H), -1e30, dtype=sc.dtype).scatter_reduce(0, rows.unsqueeze(1).expand_as(sc), sc.detach(),
                                                                      reduce="amax")
        ex = torch.exp(sc - mx[rows])
        den = torch.zeros((n, H), dtype=sc.dtype).index_add(0, rows, ex)
        alpha = ex / (den[rows] + 1e-8)
        out = torch.zeros((n, H, dv), dtype=V.dtype).index_add(0, rows, alpha.unsqueeze(-1) * v)
        return out.reshape(n, H * dv)

    def hub_lists(self, row_begin, row_end, rp_stride, n_dst, num_edges):
        """Same chunking policy as HipBackend.hub_lists (plan.hub_policy / build_hub_lists are plain torch)."""
        from tf_geometric_amd.plan import build_hub_lists, hub_policy
        thr, chunk = hub_policy(num_edges, n_dst)
        idx = torch.arange(n_dst) * rp_stride
        lists = build_hub_lists(row_begin[idx], row_end[idx], thr, chunk)
        return None if lists is None else (thr,) + lists

    def segment_reduce(self, row_begin, row_end, rp_stride, col, w, n_dst, x, out, op, act=0, accumulate=False,
                       self_coef=None, bias=None, mean_count=None, hub=None, split=None):
        rb, re, c, wv, xv = _np(row_begin), _np(row_end), _np(col), _np(w), _np(x).astype(np.float64)
        o = _np(out)
        for r in range(n_dst):
            s, e = int(rb[r * rp_stride]), int(re[r * rp_stride])
            msg = xv[c[s:e]]
            if wv is not None:
                msg = msg * wv[s:e, None].astype(np.float64)
            if op == MAX:
                acc = msg.max(axis=0) if e > s else np.full(xv.shape[1], FLT_LOWEST, np.float64)
                if accumulate:
                    acc = np.maximum(acc, o[r])
                if self_coef is not None:
                    acc = np.maximum(acc, float(self_coef[r]) * xv[r])
            else:
                acc = msg.sum(axis=0)
                if accumulate:
                    acc = acc + o[r]
                if self_coef is not None:
                    acc = acc + float(self_coef[r]) * xv[r]
                if op == MEAN:
                    cnt = int(mean_count[r]) if mean_count is not None else (e - s)
                    acc = acc / max(cnt, 1)
            if bias is not None:
                acc = acc + _np(bias)
            if act == 1:
                acc = np.maximum(acc, 0)
            o[r] = acc.astype(np.float32)
        return out

    def weight_sum(self, row_ptr, w, n, diag):
        rp, wv = _np(row_ptr), _np(w)
        deg = np.array([(wv[rp[r]:rp[r + 1]].astype(np.float64).sum() if wv is not None else rp[r + 1] - rp[r]) + diag
                        for r in range(n)], dtype=np.float32)
        return torch.from_numpy(deg)

    def gcn_norm_edges(self, row_ptr, col, w, n, row_deg, mode, fill, add_self_loop, renorm, col_deg=None):
        rp, c, wv, deg = _np(row_ptr), _np(col), _np(w), _np(row_deg).astype(np.float64)
        cdeg = deg if col_deg is None else _np(col_deg).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = np.power(cdeg, -0.5 if mode == 0 else -1.0)
        pc = np.where(np.isfinite(pc), pc, 0.0)
        E = c.shape[0]
        wv = np.ones(E, np.float64) if wv is None else wv.astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            p = np.power(deg, -0.5 if mode == 0 else -1.0)
        p = np.where(np.isfinite(p), p, 0.0)
        rows = np.repeat(np.arange(n), np.diff(rp))
        if mode == 0:
            w_out = p[rows] * wv * pc[c]
            sc = (p[:n] * fill * pc[:n]) if renorm else np.full(n, fill)
        elif mode == 1:
            w_out = p[rows] * wv
            sc = p[:n] * fill
        else:
            w_out = wv * p[c]
            sc = fill * p[:n]
        if not add_self_loop:
            sc = np.zeros(n)
        return torch.from_numpy(w_out.astype(np.float32)), torch.from_numpy(sc.astype(np.float32))

    def gemm_bias_act(self, a, b, bias=None, act=0, out=None):
        res = _np(a).astype(np.float64) @ _np(self.f32(b)).astype(np.float64)
        if bias is not None:
            res = res + _np(bias)
        if act == 1:
            res = np.maximum(res, 0)
        res = torch.from_numpy(res.astype(np.float32))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def gat_pass(self, row_begin, row_end, rp_stride, col, n_dst, Q, K, V, num_heads, state_acc, state_ml,
                 skip_longer_than=0, part_row=None):
        rb, re, c = _np(row_begin), _np(row_end), _np(col)
        q, k, v = _np(Q).astype(np.float64), _np(K).astype(np.float64), _np(V).astype(np.float64)
        pr = None if part_row is None else _np(part_row)
        H = num_heads
        d, dv = q.shape[1] // H, v.shape[1] // H
        acc, ml = _np(state_acc), _np(state_ml)
        for p in range(n_dst):
            s, e = int(rb[p * rp_stride]), int(re[p * rp_stride])
            if skip_longer_than > 0 and e - s > skip_longer_than:
                acc[p], ml[p] = np.nan, np.nan            # must never be read: its chunks are merged instead
                continue
            r = p if pr is None else int(pr[p])
            cols = c[s:e]
            for h in range(H):
                if e == s:
                    acc[p, h * dv:(h + 1) * dv] = 0
                    ml[p, 2 * h], ml[p, 2 * h + 1] = FLT_LOWEST, 0.0
                    continue
                sc = (k[cols, h * d:(h + 1) * d] @ q[r, h * d:(h + 1) * d]) / np.sqrt(d)
                m = sc.max()
                pe = np.exp(sc - m)
                acc[p, h * dv:(h + 1) * dv] = (pe[:, None] * v[cols, h * dv:(h + 1) * dv]).sum(0)
                ml[p, 2 * h], ml[p, 2 * h + 1] = m, pe.sum()

    def gat_merge_parts(self, Q, K, V, num_heads, n_dst, state_acc, state_ml, part_ptr, part_idx, bias, act, out):
        q, k, v = _np(Q).astype(np.float64), _np(K).astype(np.float64), _np(V).astype(np.float64)
        acc, ml = _np(state_acc).astype(np.float64), _np(state_ml).astype(np.float64)
        pp, pi = _np(part_ptr), _np(part_idx)
        H = num_heads
        d, dv = q.shape[1] // H, v.shape[1] // H
        res = np.zeros((n_dst, v.shape[1]))
        for r in range(n_dst):
            parts = pi[pp[r]:pp[r + 1]]
            for h in range(H):
                s_self = (q[r, h * d:(h + 1) * d] @ k[r, h * d:(h + 1) * d]) / np.sqrt(d)
                live = [p for p in parts if ml[p, 2 * h + 1] > 0]
                M = max([s_self] + [ml[p, 2 * h] for p in live])
                L_ = np.exp(s_self - M)
                O = L_ * v[r, h * dv:(h + 1) * dv]
                for p in live:
                    cf = np.exp(ml[p, 2 * h] - M)
                    L_ += ml[p, 2 * h + 1] * cf
                    O = O + acc[p, h * dv:(h + 1) * dv] * cf
                res[r, h * dv:(h + 1) * dv] = O / (L_ + 1e-8)
        assert np.isfinite(res).all(), "a skipped (hub) state row was merged"
        if bias is not None:
            res = res + _np(bias)
        if act == 1:
            res = np.maximum(res, 0)
        out.copy_(torch.from_numpy(res.astype(np.float32)))
        return out

    def gat_merge(self, Q, K, V, num_heads, n_dst, state_acc, state_ml, n_passes, bias, act, out):
        q, k, v = _np(Q).astype(np.float64), _np(K).astype(np.float64), _np(V).astype(np.float64)
        acc, ml = _np(state_acc).astype(np.float64), _np(state_ml).astype(np.float64)
        H = num_heads
        d, dv = q.shape[1] // H, v.shape[1] // H
        o = _np(out)
        for r in range(n_dst):
            for h in range(H):
                s_self = (q[r, h * d:(h + 1) * d] @ k[r, h * d:(h + 1) * d]) / np.sqrt(d)
                ms = [ml[t * n_dst + r, 2 * h] for t in range(n_passes)]
                ls = [ml[t * n_dst + r, 2 * h + 1] for t in range(n_passes)]
                M = max([s_self] + [m for m, l in zip(ms, ls) if l > 0])
                L_ = np.exp(s_self - M)
                O = L_ * v[r, h * dv:(h + 1) * dv]
                for t in range(n_passes):
                    if ls[t] > 0:
                        cf = np.exp(ms[t] - M)
                        L_ += ls[t] * cf
                        O = O + acc[t * n_dst + r, h * dv:(h + 1) * dv] * cf
                o[r, h * dv:(h + 1) * dv] = O / (L_ + 1e-8)
        res = o.astype(np.float64)
        if bias is not None:
            res = res + _np(bias)
        if act == 1:
            res = np.maximum(res, 0)
        out.copy_(torch.from_numpy(res.astype(np.float32)))
        return out
