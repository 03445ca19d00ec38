# coding=utf-8
"""Transformer-style multi-head GAT on the HIP backend — functional mirror of tf_geometric/nn/conv/gat.py.

Three MFMA GEMMs (Q, K with bias+activation fused; V), then ONE fused launch per layer that walks each
destination row once: per-head score <Q[row], K[col]>/sqrt(d), online edge-softmax and the weighted sum of V.
The reference's gathered [E', A] Q/K tensors (gat.py:56,65), the H-fold virtual graph (:73-76) and the score
vector (:79) are never materialised; the N self-loop edges of add_self_loop_edge (:43) are implicit.
"""
import ctypes
import math
import os

import torch

from ... import _lib as L
from ...activations import resolve as _resolve_act, relu as relu_act
from ...plan import CsrPlan, gemm_bias_act, gather_friendly_empty
from ...sparse import sparse_features, sparse_dense_matmul
from ... import autograd as AG


def _linear(x, kernel, bias, activation):
    act, post = _resolve_act(activation)
    h = gemm_bias_act(x, kernel, bias=bias, act=act)
    return post(h) if post is not None else h


def _values(x, wv):
    """V = x @ W, rows gathered per edge by the attention kernel: written at a line-friendly stride (U = 44 -> 48)."""
    return gemm_bias_act(x, wv, out=gather_friendly_empty(int(x.shape[0]), int(wv.shape[1]), x.device))


def _project_qkv(x, wq, bq, qact, wk, bk, kact, wv):
    """Q = qact(x@Wq+bq), K = kact(x@Wk+bk), V = x@W (gat.py:52-70): three launches of the MFMA GEMM.
    (One fused x @ [Wq | Wk | W] pass with a column-limited activation — tfgx_gemm_bias_act_cols_f32 — was measured
    at products shape, H=8/A=8/U=64: 8.1 ms per layer vs 7.6 ms for the three GEMMs, because K and V then share
    320-byte rows that straddle more 128-byte lines in the attention kernel.  A second attempt with rows padded to
    whole lines ([Q | K | pad | V], V line-aligned, K in a line of its own) kept the attention kernel at 3 lines per
    edge but the layer still took 7.55-7.58 ms: each of the three GEMMs is already bound by streaming x once, and the
    fused one is a low-efficiency N = 96 tile shape.  Not used.)
    What IS fused: Q and K when they share the activation — one x @ [Wq | Wk] pass instead of two (both are narrow,
    HBM-bound reads of x); K rows stay inside one line (A <= 16) or line-aligned (A % 32 == 0)."""
    qa, qpost = _resolve_act(qact)
    ka, kpost = _resolve_act(kact)
    A = int(wq.shape[1])
    xs = sparse_features(x)
    if xs is not None:      # sparse node features: the three projections are sparse_dense_matmuls (gat.py:49-68)
        def sp(wm, bm, code, post):
            h = sparse_dense_matmul(xs, wm, bias=bm, act=code)
            return post(h) if post is not None else h
        return sp(wq, bq, qa, qpost), sp(wk, bk, ka, kpost), sparse_dense_matmul(xs, wv)
    if (qa == ka and qpost is None and kpost is None and (bq is None) == (bk is None) and int(wk.shape[1]) == A and
            (A <= 16 or A % 32 == 0)):
        dev = x.device
        w_qk = torch.cat([L.as_f32(wq, dev), L.as_f32(wk, dev)], dim=1).contiguous()
        b_qk = None if bq is None else torch.cat([L.as_f32(bq, dev).reshape(-1), L.as_f32(bk, dev).reshape(-1)])
        qk = gemm_bias_act(x, w_qk, bias=b_qk, act=qa)
        # K rows narrower than a 128-byte line are gathered per edge: inside [Q | K] a line holds half as many of them, and
        # the attention kernel (no longer hidden behind its own vector-ALU work) ran 3.62 instead of 3.09 ms at the Reddit
        # shape (tools/gat_layer_attention_probe.py) — a copy of [n, A] floats costs ~10 us
        K = qk[:, A:].contiguous() if A <= 16 else qk[:, A:]
        return qk[:, :A], K, _values(x, wv)
    return _linear(x, wq, bq, qact), _linear(x, wk, bk, kact), _values(x, wv)


def gat_args(Q, K, V, num_heads, n_dst, col, add_self_loop=True, bias=None, act=L.ACT_NONE, out=None, scale_d=None):
    """Fill a tfgx_gat_args for Q:[n_dst,A] K:[n_src,A] V:[n_src,W]; returns (args, out, keep-alive tuple)."""
    Q, ldq = L.row_major_2d(Q)
    K, ldk = L.row_major_2d(K)
    V, ldv = L.row_major_2d(V)
    A, W = int(Q.shape[1]), int(V.shape[1])
    if A % num_heads or W % num_heads:
        raise ValueError("attention_units ({}) and value width ({}) must be divisible by num_heads ({})"
                         .format(A, W, num_heads))
    if out is None:
        out = torch.empty((n_dst, W), dtype=torch.float32, device=V.device)
    a = L.GatArgs()
    a.col = col.data_ptr()
    a.n_dst = n_dst
    a.q, a.ldq = Q.data_ptr(), ldq
    a.k, a.ldk = K.data_ptr(), ldk
    a.v, a.ldv = V.data_ptr(), ldv
    a.out, a.ldo = out.data_ptr(), max(W, 1)
    a.H, a.d, a.dv = num_heads, A // num_heads, W // num_heads
    a.add_self_loop = 1 if add_self_loop else 0
    # gat.py:78  sqrt(shape(Q_)[-1]); scale_d: the reference's per-head width when Q / K arrive zero-padded per head
    a.scale = math.sqrt(float(A // num_heads if scale_d is None else scale_d))
    a.act = act
    a.bias = 0 if bias is None else bias.data_ptr()
    return a, out, (Q, K, V)


_SEED_STATE = {}          # (device index, torch seed) -> int64[1] device tensor: the seed stream hipGraph-captured steps draw from
_SEED_STRIDE = -7046029254386353131      # 0x9E3779B97F4A7C15 as int64 (odd: the stream visits all 2^64 values)


def _seed_state(device):
    """Created EAGERLY (never inside a capture: a captured fill would reset it on every replay), from torch's host generator."""
    # one stream per (device, torch seed): torch.manual_seed(s) before a capture makes the captured masks repeatable; streams
    # of earlier seeds stay alive (a captured graph keeps reading the one it was captured with)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.initial_seed())
    st = _SEED_STATE.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise L.TfgxError("attention dropout inside a hipGraph capture needs the device seed stream to exist before the "
                              "capture starts: run the step once eagerly first (CapturedTrainStep's warm-up does)")
        # a private generator keyed by torch's seed: repeatable under torch.manual_seed, and the GLOBAL host stream — which
        # the eager by-value seeds are drawn from — is not advanced by creating the device stream
        g = torch.Generator()
        g.manual_seed(torch.initial_seed() % (2 ** 63))
        hi, lo = torch.randint(0, 2 ** 31 - 1, (2,), generator=g).tolist()
        st = torch.tensor([(int(hi) << 32) | int(lo)], dtype=torch.int64, device=device)
        _SEED_STATE[key] = st
    return st


def new_drop_seed(device=None):
    """A fresh 64-bit dropout seed.  Eager launches: a host integer from torch's host generator (torch.manual_seed makes
    training runs repeatable), passed by value.  While the current stream is being CAPTURED into a hipGraph a by-value seed
    would be frozen into the graph — every replay would drop the same edges — so the seed is a DEVICE tensor there: the
    captured sequence advances the per-device seed stream in place (an add kernel, replayed with the step) and snapshots it
    for this layer call; forward and backward kernels read the snapshot (tfgx_gat_args.drop_seed_dev)."""
    if device is not None:
        state = _seed_state(device)              # exists before any capture (the eager warm-up steps create it)
        if torch.cuda.is_current_stream_capturing():
            state.add_(_SEED_STRIDE)
            return state.clone()
    hi, lo = torch.randint(0, 2 ** 31 - 1, (2,)).tolist()
    return (int(hi) << 32) | int(lo)


def _set_drop(a, drop_rate, drop_seed, self_base):
    """drop_seed: host int, or an int64[1] device tensor (see new_drop_seed)."""
    a.drop_rate, a.drop_self_base = float(drop_rate), int(self_base)
    if isinstance(drop_seed, torch.Tensor):
        a.drop_seed, a.drop_seed_dev = 0, drop_seed.data_ptr()
    else:
        a.drop_seed = int(drop_seed) & 0xFFFFFFFFFFFFFFFF


SOURCE_BLOCKS = None      # developer A/B: None = the policy below, 0 / 1 = always one pass, k >= 2 = always k source blocks
SOURCE_BLOCK_BYTES = 6 << 20          # K | V rows gathered per pass.  Reddit shape (67 MB table), tools/gat_kb_sweep.py: KB = 4 / 6 / 8 /
SOURCE_BLOCK_MIN_EDGES = 32           # 10 / 12 / 14 / 16 / 20 / 24 -> 4.14 / 3.50 / 2.98 / 2.77 / 2.77 / 2.80 / 2.83 / 2.93 / 3.09 ms (the
                                      # optimum was KB = 8 while the kernel was bound by its own vector-ALU work); and at least this
                                      # many edges per (row, block) on average
SOURCE_BLOCK_STATS = {"launches": 0}  # diagnostics (tests assert the route taken)
DESTINATION_BLOCKS = None             # developer A/B: the number of destination blocks of the backward's source pass (None = the policy)


def source_block_count(plan, A, W):
    """How many source blocks the fused attention runs in (1 = one pass).  Dense graphs only: the K | V table
    (n_src x (A + W) floats) must be several blocks large and every (row, block) must still hold a few dozen edges —
    Reddit shape (233 k nodes, 489 in-edges each, A = 8, W = 64): 11 blocks; products shape (51 in-edges): 1."""
    if SOURCE_BLOCKS is not None:
        return max(int(SOURCE_BLOCKS), 1)
    if plan.n_dst == 0 or plan.num_edges == 0:
        return 1
    table = plan.n_src * (int(A) + int(W)) * 4
    kb = min(int(round(table / float(SOURCE_BLOCK_BYTES))), int(plan.num_edges / float(plan.n_dst) / SOURCE_BLOCK_MIN_EDGES), 16)
    return kb if kb >= 2 else 1


def _gat_attention_source_blocks(plan, blocks, KB, Q, K, V, num_heads, add_self_loop, bias, act, stats_ml, scale_d, qsums=None):
    """KB chained launches of tfgx_gat_fused_f32, launch b over the edges whose source lies in block b: the raw online-softmax
    state of every row is handed from launch to launch (tfgx_gat_args.state_in_*), the last launch appends the self-loop
    edge and finishes the rows.  Every launch gathers K / V rows of ONE block, which the L2 of every XCD then serves
    (Reddit shape: 5.8 -> 3.5 ms).  The result differs from the one-pass kernel only by the order in which a row's edges
    enter its softmax sums (by source block, original order inside a block)."""
    lib = L.require_gpu()
    rpk, col_k = blocks
    n, W = plan.n_dst, int(V.shape[1])
    dev = V.device
    bufs = [(torch.empty((n, W), dtype=torch.float32, device=dev), torch.empty((n, 2 * num_heads), dtype=torch.float32, device=dev))
            for _ in range(2 if KB > 2 else 1)]
    # the query gradient's sums (tfgx_gat_args.qgrad_t) travel with the state
    tbufs = [(torch.empty((n, W), dtype=torch.float32, device=dev), torch.empty((n, num_heads), dtype=torch.float32, device=dev))
             for _ in bufs] if qsums is not None else None
    out = None
    prev = prev_t = None
    for b in range(KB):
        last = b == KB - 1
        a, o, keep = gat_args(Q, K, V, num_heads, n, col_k, add_self_loop if last else False, bias if last else None,
                              act if last else L.ACT_NONE, scale_d=scale_d, out=None if last else bufs[b % len(bufs)][0])
        a.row_begin, a.row_end, a.rp_stride = rpk[b:].data_ptr(), rpk[b + 1:].data_ptr(), KB
        if prev is not None:
            a.state_in_acc, a.state_in_ml = prev[0].data_ptr(), prev[1].data_ptr()
            if qsums is not None:
                a.state_in_t, a.state_in_s = prev_t[0].data_ptr(), prev_t[1].data_ptr()
        if last:
            out = o
            if stats_ml is not None:
                a.stats_ml = stats_ml.data_ptr()
            if qsums is not None:
                a.qgrad_t, a.qgrad_s = qsums[0].data_ptr(), qsums[1].data_ptr()
        else:
            cur = bufs[b % len(bufs)]
            a.state_acc, a.state_ml = cur[0].data_ptr(), cur[1].data_ptr()
            prev = cur
            if qsums is not None:
                prev_t = tbufs[b % len(bufs)]
                a.state_t, a.state_s = prev_t[0].data_ptr(), prev_t[1].data_ptr()
        L.check(lib.tfgx_gat_fused_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_gat_fused_f32")
    SOURCE_BLOCK_STATS["launches"] += KB
    return out


QUERY_GRAD_SUMS = os.environ.get("TFGX_GAT_QUERY_SUMS", "1") != "0"   # developer A/B: 0 = dQ always from the backward's destination pass


def query_sums_apply(plan, Q, V, num_heads, drop_rate=0.0):
    """True when the training forward can accumulate the query gradient's sums (tfgx_gat_args.qgrad_t: one attention unit
    per head, 4-column lanes, no hub lists) — the backward then needs no destination pass."""
    A, W = int(Q.shape[1]), int(V.shape[1])
    if not QUERY_GRAD_SUMS or A != num_heads or W % num_heads or (W // num_heads) % 4 or plan.n_dst == 0:
        return False
    if drop_rate <= 0.0 and plan.hub_info() is not None:
        return False
    V2, ldv = L.row_major_2d(V)
    return ldv % 4 == 0 and V2.data_ptr() % 16 == 0


def gat_attention(plan, Q, K, V, num_heads, add_self_loop=True, bias=None, act=L.ACT_NONE, stats_ml=None,
                  drop_rate=0.0, drop_seed=0, scale_d=None, qsums=None):
    """Fused SDDMM + edge softmax + SpMM over `plan` (tfgx_gat_fused_f32). Q:[n_dst,A] K:[n_src,A] V:[n_src,W].
    Destinations with very many in-edges (plan.hub_info()) are processed chunk-wise and merged.
    drop_rate > 0 (training): the softmax weights are dropped / rescaled inside the kernel (gat.py:85); the keep
    decision is a function of (drop_seed, CSR position, head), positions [E, E+n) being the appended self-loops."""
    lib = L.require_gpu()
    if drop_rate <= 0.0 and plan.row_order() is None and plan.hub_info() is None:
        KB = source_block_count(plan, int(Q.shape[1]), int(V.shape[1]))
        blocks = plan.source_blocks(KB) if KB >= 2 else None
        if blocks is not None:
            return _gat_attention_source_blocks(plan, blocks, KB, Q, K, V, num_heads, add_self_loop, bias, act, stats_ml, scale_d,
                                                qsums)
    a, out, keep = gat_args(Q, K, V, num_heads, plan.n_dst, plan.col, add_self_loop, bias, act, scale_d=scale_d)
    a.row_ptr = plan.row_ptr.data_ptr()
    order = plan.row_order()                 # skewed graphs: rows of similar length share a wave
    if order is not None:
        a.row_order = order.data_ptr()
    if stats_ml is not None:
        a.stats_ml = stats_ml.data_ptr()      # (m, l) per row and head, kept for the backward pass
    if qsums is not None:                     # (T, S): see query_sums_apply
        a.qgrad_t, a.qgrad_s = qsums[0].data_ptr(), qsums[1].data_ptr()
    if drop_rate > 0.0:
        _set_drop(a, drop_rate, drop_seed, plan.num_edges)
    hub = plan.hub_info() if drop_rate <= 0.0 else None      # the chunk-merge path has no dropout; long rows run inline
    if hub is not None:
        hub_rows, chunk_ptr, chunk_begin, chunk_end, chunk_row = hub
        W, nc = int(keep[2].shape[1]), int(chunk_begin.shape[0])
        s_acc = torch.empty((nc, W), dtype=torch.float32, device=out.device)
        s_ml = torch.empty((nc, 2 * num_heads), dtype=torch.float32, device=out.device)
        a.hub_threshold = plan.hub_threshold
        a.hub_rows, a.hub_chunk_ptr = hub_rows.data_ptr(), chunk_ptr.data_ptr()
        a.hub_chunk_begin, a.hub_chunk_end = chunk_begin.data_ptr(), chunk_end.data_ptr()
        a.hub_chunk_row = chunk_row.data_ptr()
        a.n_hub_rows, a.n_hub_chunks = int(hub_rows.shape[0]), nc
        a.hub_scratch_acc, a.hub_scratch_ml = s_acc.data_ptr(), s_ml.data_ptr()
    L.check(lib.tfgx_gat_fused_f32(ctypes.byref(a), L.stream_ptr()), "tfgx_gat_fused_f32")
    return out


_FAST_D = (1, 2, 4, 8, 16, 32)      # attention units per head the fused kernels are instantiated for


def _pow2_ceil(v):
    return 1 << max(int(v) - 1, 0).bit_length()


def _kernel_widths(d, dv, num_heads):
    """(d', dv'): per-head widths the fast attention kernels (forward and both backward passes) accept — d in
    {1, 2, 4, 8, 16, 32}; dv a multiple of 4 whose lane count dv / 4 is a power of two (several heads) — or the widths
    themselves when they already qualify / cannot be helped.  Zero columns change neither <Q, K> nor the real output
    columns; without them the layer falls to the one-lane-per-(row, head) kernels (10-100 x slower on large graphs)."""
    d2 = d if (d in _FAST_D or d > 32) else min(v for v in _FAST_D if v >= d)
    dv2 = dv
    if num_heads > 1:
        lanes = -(-dv // 4)
        if dv % 4 != 0 or (lanes & (lanes - 1)) != 0:
            cand = 4 * _pow2_ceil(lanes)
            dv2 = cand if cand // 4 <= 64 else dv
    return d2, dv2


def _pad_heads(t, num_heads, w_new):
    """[n, H * w] -> [n, H * w_new], every head's block zero-padded on the right (differentiable)."""
    n, w = int(t.shape[0]), int(t.shape[1]) // num_heads
    if w_new == w:
        return t
    return torch.nn.functional.pad(t.reshape(n, num_heads, w), (0, w_new - w)).reshape(n, num_heads * w_new)


def _unpad_heads(t, num_heads, w):
    n, w_pad = int(t.shape[0]), int(t.shape[1]) // num_heads
    if w_pad == w:
        return t
    return t.reshape(n, num_heads, w_pad)[:, :, :w].reshape(n, num_heads * w)


def _gat_train(x, plan, wq, bq, qact, wk, bk, kact, kernel, bias, activation, num_heads, split_value_heads,
               drop_rate=0.0):
    """Differentiable route (autograd.py): same math, un-fused epilogues."""
    xs = sparse_features(x)

    def lin(w, b, actv):
        code, post = _resolve_act(actv)
        if xs is not None:
            return AG.apply_activation(sparse_dense_matmul(xs, w, bias=b, act=code), L.ACT_NONE, post)
        return AG.apply_activation(AG.linear(x, w, b, code), L.ACT_NONE, post)
    qa, qpost = _resolve_act(qact)
    ka, kpost = _resolve_act(kact)
    if (xs is None and qa == ka and qpost is None and kpost is None and (bq is None) == (bk is None)
            and int(wq.shape[1]) == int(wk.shape[1])):
        # one operator for the three projections: x is read twice in the forward and ONCE in the backward (autograd._ProjectQKV)
        Q, K, V = AG.project_qkv(x, wq, bq, wk, bk, kernel, qa)
    else:
        Q, K, V = lin(wq, bq, qact), lin(wk, bk, kact), (sparse_dense_matmul(xs, kernel) if xs is not None
                                                          else AG.linear(x, kernel, gathered=True))
    d, dv = int(Q.shape[1]) // num_heads, int(V.shape[1]) // num_heads
    d2, dv2 = _kernel_widths(d, dv, num_heads)
    seed = new_drop_seed(V.device) if drop_rate > 0.0 else 0
    Qp, Kp = _pad_heads(Q, num_heads, d2), _pad_heads(K, num_heads, d2)
    if num_heads == 1 and dv > 256:
        # one head wider than the backward kernels' 64 lanes x 4 columns: the attention weights do not depend on the
        # value columns, so the layer is exactly the concatenation of the same attention over 256-column blocks of V
        # (same dropout seed -> same kept edges); autograd sums the blocks' d/dQ, d/dK
        h = torch.cat([AG.gat_attention(plan, Qp, Kp, V[:, c0:c0 + 256], 1, drop_rate=drop_rate, drop_seed=seed, scale_d=d)
                       for c0 in range(0, dv, 256)], dim=1)
    else:
        h = AG.gat_attention(plan, Qp, Kp, _pad_heads(V, num_heads, dv2), num_heads, drop_rate=drop_rate,
                             drop_seed=seed, scale_d=d)
        h = _unpad_heads(h, num_heads, dv)
    if not split_value_heads:
        U = int(V.shape[1]) // num_heads
        h = h.view(h.shape[0], num_heads, U).sum(1) / num_heads
    if bias is not None:
        h = AG.bias_add(h, bias)
    code, post = _resolve_act(activation)
    return AG.apply_activation(h, code, post)


def gat(x, edge_index,
        query_kernel, query_bias, query_activation,
        key_kernel, key_bias, key_activation,
        kernel, bias=None, activation=None, num_heads=1,
        split_value_heads=True, edge_drop_rate=0.0, training=False, cache=None):
    """
    Functional GAT (reference: gat.py:13-122; same arguments, plus an optional `cache` dict for the CSR plan).

    :param x: [num_nodes, num_features]
    :param edge_index: [2, num_edges]
    :param split_value_heads: True -> V is split into heads and the heads are concatenated (:112);
        False -> kernel is [F, units*num_heads] and the heads are averaged (:114).
    :return: [num_nodes, num_output_features]
    """
    lib = L.require_gpu()
    drop = float(edge_drop_rate) if training else 0.0           # SparseMatrix.dropout(rate, training) (:85)
    if not 0.0 <= drop < 1.0:
        raise Exception("edge_drop_rate must be in [0, 1)")
    U_out = int(kernel.shape[1])
    if num_heads == 1 and U_out % 4 != 0:
        # single-head output layers have odd widths (demo/demo_gat.py:23: 41 Reddit classes): value rows of U floats are
        # neither 16-byte aligned nor a whole number of vector lanes, and the attention kernels (forward AND both backward
        # passes) fall back to scalar lanes — 9.3 ms instead of 4.6 ms at the Reddit shape.  Up to three ZERO columns
        # appended to the value kernel (and bias) make the width a multiple of four: the extra output columns are
        # act(0 + 0) and are cut off again; the real columns see exactly the same arithmetic.
        pad = (-U_out) % 4
        kf = L.as_f32(kernel)
        kernel_p = torch.cat([kf, kf.new_zeros((int(kf.shape[0]), pad))], dim=1)
        bias_p = None
        if bias is not None:
            bf = L.as_f32(bias)
            bias_p = torch.cat([bf, bf.new_zeros(pad)])
        # only an ELEMENTWISE activation may see the zero columns: the fused code (None / relu) rides into the padded
        # call, a caller-supplied callable (softmax, log_softmax, a normalisation ...) runs on the unpadded result, as
        # the reference applies it (gat.py:119-120)
        act_code, act_post = _resolve_act(activation)
        h = gat(x, edge_index, query_kernel, query_bias, query_activation, key_kernel, key_bias, key_activation,
                kernel_p, bias_p, relu_act if act_code == L.ACT_RELU else None, 1, split_value_heads, edge_drop_rate,
                training, cache)
        h = h[:, :U_out].contiguous()
        return act_post(h) if act_post is not None else h
    xs = sparse_features(x)
    x = xs if xs is not None else L.as_f32(x)
    n = int(x.shape[0])
    plan = CsrPlan.from_cache(edge_index, n, n, cache)
    if drop > 0.0 or AG.needs_grad(None if xs is not None else x, query_kernel, query_bias, key_kernel, key_bias, kernel,
                                   bias):
        return _gat_train(x, plan, query_kernel, query_bias, query_activation, key_kernel, key_bias, key_activation,
                          kernel, bias, activation, num_heads, split_value_heads, drop_rate=drop)
    Q, K, V = _project_qkv(x, query_kernel, query_bias, query_activation, key_kernel, key_bias, key_activation, kernel)
    act, post = _resolve_act(activation)
    bias_t = None if bias is None else L.as_f32(bias).contiguous()
    d, dv = int(Q.shape[1]) // num_heads, int(V.shape[1]) // num_heads
    d2, dv2 = _kernel_widths(d, dv, num_heads)
    Q, K = _pad_heads(Q, num_heads, d2), _pad_heads(K, num_heads, d2)
    if dv2 != dv:           # padded value heads: bias / activation cannot ride in the kernel (its columns are shifted)
        h = _unpad_heads(gat_attention(plan, Q, K, _pad_heads(V, num_heads, dv2), num_heads, True, scale_d=d), num_heads, dv)
        if not split_value_heads:
            h = h.reshape(n, num_heads, dv).sum(1) / num_heads
        if bias_t is not None:
            h = h + bias_t
        h = torch.relu(h) if act == L.ACT_RELU else h
        return post(h) if post is not None else h
    if split_value_heads:
        h = gat_attention(plan, Q, K, V, num_heads, True, bias=bias_t, act=act, scale_d=d)
    else:
        h_ = gat_attention(plan, Q, K, V, num_heads, True, scale_d=d)
        U = int(V.shape[1]) // num_heads
        h = torch.empty((n, U), dtype=torch.float32, device=x.device)
        L.check(lib.tfgx_head_mean_f32(L.ptr(h_), int(V.shape[1]), n, num_heads, U, L.ptr(bias_t), act, L.ptr(h),
                                       max(U, 1), L.stream_ptr()), "tfgx_head_mean_f32")
    return post(h) if post is not None else h
