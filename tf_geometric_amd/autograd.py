# coding=utf-8
"""Differentiable wrappers of the hot-path kernels (the backward pass).

The reference trains with tf.GradientTape over TensorFlow ops (demo/demo_gcn.py:68-77); here the host tensor
library is PyTorch, so the same role is played by torch.autograd.Function objects whose forward AND backward are
C-ABI kernel launches:

  aggregate (sum / mean, weighted)  backward wrt x  = the forward kernel on the transposed (CSR-by-source) plan
                                    backward wrt w  = tfgx_sddmm_f32
  aggregate (max)                   training forward saves (max, tie count, arg position); backward = per-edge winner masks +
                                    one gather per edge (tfgx_segment_max_backward_mask_f32, bit-reproducible; ties walked
                                    exactly, TF semantics); MAX_GRADIENT_MODE selects the atomics ("push") or two-gather ("pull") forms
  gat_attention                     tfgx_gat_backward_dst_f32 (dQ) + tfgx_gat_backward_src_f32 (dK, dV)
  linear (x @ W + b, relu)          forward = tfgx_gemm_bias_act_f32; d/dx = the same kernel on W^T (tfgx_transpose_f32);
                                    d/dW, d/db = tfgx_gemm_tn_f32 (MFMA reduction over the node dimension)
  segment_softmax                   forward = tfgx_edge_softmax_f32; backward = out * (g - segsum(out * g)) on the segment kernel

The functional API (nn/conv/*.py) routes through these only when torch.is_grad_enabled() and an input requires
grad; inference keeps the fused single-launch paths.
"""
import ctypes
import math

import torch

from . import _lib as L
from .plan import (segment_reduce, can_track, gemm_bias_act, gemm_tn, transpose, SplitRows, gather_friendly_empty,
                   gather_friendly_copy, aggregate_gemm, aggregate_gemm_applies, column_sums)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def _transposed(plan):
    """(transposed plan, t2d) where t2d[j] = forward-CSR position of the edge at transposed-CSR position j."""
    if getattr(plan, "_t2d", None) is None:
        pt = plan.transposed()
        inv = torch.empty_like(plan.perm)
        inv[plan.perm.long()] = torch.arange(plan.num_edges, dtype=torch.int32, device=plan.perm.device)
        plan._t2d = inv[pt.perm.long()].contiguous()
    return plan.transposed(), plan._t2d


def _permute(attr, idx):
    lib = L.require_gpu()
    a = attr.contiguous()
    out = torch.empty_like(a)
    L.check(lib.tfgx_permute_rows_f32(L.ptr(a), L.ptr(idx), int(idx.shape[0]), 1, L.ptr(out), L.stream_ptr()),
            "tfgx_permute_rows_f32")
    return out


def _transposed_weights(plan, w_csr, t2d):
    """w_csr permuted into the transposed plan's order, memoised per (tensor storage, version) on the plan —
    GCN's normalised weights are the same tensor for every layer and every step."""
    if w_csr is None:
        return None
    key = (w_csr.data_ptr(), w_csr._version, int(w_csr.shape[0]))
    cache = plan.__dict__.setdefault("_wt_cache", {})
    hit = cache.get("w")
    if hit is not None and hit[0] == key:
        return hit[1]
    w_t = _permute(w_csr.detach(), t2d)
    cache["w"] = (key, w_t, w_csr)      # keeps w_csr alive so the data_ptr cannot be recycled
    return w_t


import os as _os
# Apply a layer-0 ReLU mask INSIDE the weight-gradient reduction (tfgx_gemm_tn_gated_f32) instead of one masked copy of the
# gradient first.  Same-box A/B at products shape: a wash (GCN step 25.84 / 26.06 vs 25.62 / 26.15 ms, mean SAGE 27.49 /
# 27.67 vs 27.52 / 27.99 ms) — the kernel reads the gate where the copy was written and re-read — so it stays off.
GATED_WEIGHT_GRADIENTS = _os.environ.get("TFGX_GATED_WGRAD", "0") != "0"
# ... and in the fused aggregate -> project layer (round 4, after the weight-gradient kernel's operand loads were rewritten):
GATED_AGGREGATE_PROJECT = _os.environ.get("TFGX_GATED_AGGPROJ", "1") != "0"


def relu_backward(g, out, into=None):
    """g where out > 0, else 0 — ONE pass (tfgx_relu_backward_f32) instead of compare + cast + multiply; g / out / into
    may be column slices of wider matrices."""
    lib = L.require_gpu()
    g2, ldg = L.row_major_2d(g)
    o2, ldo = L.row_major_2d(out)
    M, N = int(g2.shape[0]), int(g2.shape[1])
    res = torch.empty((M, N), dtype=torch.float32, device=g2.device) if into is None else into
    r2, ldr = L.row_major_2d(res)
    assert r2 is res, "relu_backward: `into` must have dense rows"
    L.check(lib.tfgx_relu_backward_f32(L.ptr(g2), ldg, L.ptr(o2), ldo, M, N, L.ptr(res), ldr, L.stream_ptr()),
            "tfgx_relu_backward_f32")
    if into is not None:
        torch.autograd.graph.increment_version(into)
    return res


def _aggregate_grad_x(plan, mean, g, w_csr, self_coef):
    """d/dx of out = (1/cnt) (sum_i w_i x[col_i] + self_coef x): the transposed plan on the same kernel; on a square
    operator the self-loop term self_coef[r] * g[r] rides in that launch's epilogue."""
    if mean:
        scaled = gather_friendly_empty(int(g.shape[0]), int(g.shape[1]), g.device)
        torch.div(g, plan.in_degree().clamp(min=1).to(g.dtype).unsqueeze(1), out=scaled)
        g = scaled
    else:
        g = gather_friendly_copy(g)        # the transposed pass GATHERS rows of g: narrow / odd widths get a padded stride
    pt, t2d = _transposed(plan)
    w_t = _transposed_weights(plan, w_csr, t2d)
    if self_coef is not None and plan.n_dst == plan.n_src:
        return segment_reduce(pt, g, L.SUM, w_csr=w_t, self_coef=self_coef.detach()), g
    gx = segment_reduce(pt, g, L.SUM, w_csr=w_t)
    if self_coef is not None:
        gx = gx + self_coef.detach().unsqueeze(1) * g
    return gx, g


class _Aggregate(torch.autograd.Function):
    """out = act( (1/cnt[r]) * ( sum_{i in row r} w[i] x[col[i]] + self_coef[r] x[r] ) + bias )   (cnt only for mean);
    bias and activation ride in the kernel's epilogue, the ReLU's backward is one masked copy of the gradient."""

    @staticmethod
    def forward(ctx, plan, mean, x, w_csr, self_coef, rows=None, bias=None, act=L.ACT_NONE):
        """`rows`: x in another source layout (plan.static_rows) — same values, same result bits."""
        ctx.plan, ctx.mean, ctx.act = plan, mean, act
        out = segment_reduce(plan, x.detach() if rows is None else rows, L.MEAN if mean else L.SUM,
                             w_csr=None if w_csr is None else w_csr.detach(),
                             self_coef=None if self_coef is None else self_coef.detach(),
                             bias=None if bias is None else bias.detach().contiguous(), act=act)
        ctx.save_for_backward(x, w_csr, self_coef, bias, out if act == L.ACT_RELU else None)
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        x, w_csr, self_coef, bias, out = ctx.saved_tensors
        g = relu_backward(g, out) if ctx.act == L.ACT_RELU else g.contiguous()
        gb = column_sums(g) if (bias is not None and ctx.needs_input_grad[6]) else None
        gx, gw, gs = _aggregate_backward(plan, ctx.mean, x, w_csr, self_coef, g, ctx.needs_input_grad[2],
                                         ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        return None, None, gx, gw, gs, None, gb, None


def _aggregate_backward(plan, mean, x, w_csr, self_coef, g, need_x, need_w, need_s):
    """(d/dx, d/dw_csr, d/dself_coef) of (1/cnt) (sum_i w_i x[col_i] + self_coef x) given g = d/d(aggregate)."""
    lib = L.require_gpu()
    gx = gw = gs = None
    need_w = need_w and w_csr is not None
    need_s = need_s and self_coef is not None
    if mean and (need_w or need_s) and not need_x:
        g = g / plan.in_degree().clamp(min=1).to(g.dtype).unsqueeze(1)
    if need_x:
        gx, g = _aggregate_grad_x(plan, mean, g, w_csr, self_coef)
    if need_w:
        gw = torch.empty_like(w_csr)
        g2, ldg = L.row_major_2d(g)
        x2, ldx = L.row_major_2d(x.detach())
        hub_w, _ = L.hub_lists(plan)           # hub destinations: their edges are walked chunk-wise
        L.check(lib.tfgx_sddmm_hub_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), plan.n_dst, L.ptr(g2), ldg, L.ptr(x2), ldx,
                                       int(x2.shape[1]), L.ptr(gw), None if hub_w is None else ctypes.byref(hub_w),
                                       L.stream_ptr()), "tfgx_sddmm_hub_f32")
    if need_s:
        gs = (x.detach() * g).sum(1)
    return gx, gw, gs


class _AggregateProject(torch.autograd.Function):
    """out = act( reduce(plan, x, w, self_coef) @ kernel + bias ) — the aggregate-then-project layers (GCN with units > F,
    the neighbour half of mean / sum GraphSAGE) in ONE forward launch (tfgx_aggregate_gemm_f32).  When the kernel needs a
    gradient the launch also writes the aggregate itself (side output: d/dkernel = aggregate^T @ g); the projection still
    reads it from LDS, so the training forward saves the GEMM's read-back of the [N, F] aggregate."""

    @staticmethod
    def forward(ctx, plan, mean, x, w_csr, self_coef, kernel, bias, act, rows=None):
        """`rows`: x in the static feature layout (plan.static_rows) — same values, same aggregate bits."""
        n, F = plan.n_dst, int(x.shape[1])
        need_agg = ctx.needs_input_grad[5]
        agg = torch.empty((n, F), dtype=torch.float32, device=x.device) if need_agg else None
        out = aggregate_gemm(plan, x.detach() if rows is None else rows, L.MEAN if mean else L.SUM, kernel.detach(),
                             w_csr=None if w_csr is None else w_csr.detach(),
                             self_coef=None if self_coef is None else self_coef.detach(),
                             bias=None if bias is None else bias.detach(), act=act, agg_out=agg)
        if out is None:
            raise L.TfgxError("_AggregateProject: the fused launch declined (ask plan.aggregate_gemm_applies first)")
        ctx.plan, ctx.mean, ctx.act = plan, mean, act
        ctx.save_for_backward(x, w_csr, self_coef, kernel, bias, agg, out if act == L.ACT_RELU else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w_csr, self_coef, kernel, bias, agg, out = ctx.saved_tensors
        need = ctx.needs_input_grad
        want_b = bias is not None and need[6]
        # layer 0 (nothing upstream wants a gradient): the ReLU mask is applied INSIDE the weight-gradient reduction (gate = the
        # layer's output) and the masked gradient is never written — one 3 x [N, units] pass less (1.45 ms at products shape;
        # the gated kernel costs 0.05 ms more than the plain one since its operands moved to raw buffer loads)
        gated = (ctx.act == L.ACT_RELU and agg is not None and (need[5] or want_b)
                 and not (need[2] or need[3] or need[4]) and GATED_AGGREGATE_PROJECT)
        if not gated:
            g = relu_backward(g, out) if ctx.act == L.ACT_RELU else g.contiguous()
        gk = gb = None
        if need[5] or want_b:
            if agg is not None:
                gk, gb = gemm_tn(agg, g.contiguous() if gated else g, want_bias=want_b, gate=out if gated else None)
                if not need[5]:
                    gk = None
            else:
                gb = column_sums(g)
        gx = gw = gs = None
        if need[2] or need[3] or need[4]:
            g_agg = gemm_bias_act(g, transpose(kernel.detach()))          # d/d(aggregate) = g @ kernel^T
            gx, gw, gs = _aggregate_backward(ctx.plan, ctx.mean, x, w_csr, self_coef, g_agg, need[2], need[3], need[4])
        return None, None, gx, gw, gs, gk, gb, None, None


def aggregate_project(plan, x, op, kernel, w_csr=None, self_coef=None, bias=None, act=L.ACT_NONE, rows=None):
    """Differentiable act(reduce(plan, x) @ kernel + bias) on the fused launch, or None when it does not take the shape
    (the caller then composes aggregate + linear).  rows: x's static layout (plan.static_rows) or None / x itself."""
    rows = rows if isinstance(rows, SplitRows) else None
    if op not in (L.SUM, L.MEAN) or not aggregate_gemm_applies(x if rows is None else rows, kernel, op):
        return None
    return _AggregateProject.apply(plan, op == L.MEAN, x, w_csr, self_coef, L.as_f32(kernel),
                                   None if bias is None else L.as_f32(bias), act, rows)


class _SageWide(torch.autograd.Function):
    """h = act([x @ ks | reduce(w * x[col]) @ kn] + bias): mean / sum GraphSAGE, concat form, aggregation first (ku >= F:
    nn/conv/graph_sage.py:34-58 with demo_graph_sage.py:29-30's units=256).  The neighbour half is ONE launch
    (tfgx_aggregate_gemm_f32 straight into its half of h, aggregate written beside it for d/dkn), the self half the GEMM."""

    @staticmethod
    def forward(ctx, plan, mean, x, ks, kn, w_csr, bias, act, rows=None):
        n, F, na, nb = int(x.shape[0]), int(x.shape[1]), int(ks.shape[1]), int(kn.shape[1])
        h = torch.empty((n, na + nb), dtype=torch.float32, device=x.device)
        bd = None if bias is None else bias.detach().contiguous()
        agg = torch.empty((n, F), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[4] else None
        got = aggregate_gemm(plan, x.detach() if rows is None else rows, L.MEAN if mean else L.SUM, kn.detach(),
                             w_csr=None if w_csr is None else w_csr.detach(),
                             bias=None if bd is None else bd[na:].contiguous(), act=act, out=h[:, na:], agg_out=agg)
        if got is None:
            raise L.TfgxError("_SageWide: the fused launch declined (ask plan.aggregate_gemm_applies first)")
        gemm_bias_act(x.detach(), ks.detach(), bias=None if bd is None else bd[:na], act=act, out=h[:, :na])
        ctx.plan, ctx.mean, ctx.act, ctx.na = plan, mean, act, na
        ctx.save_for_backward(x, ks, kn, w_csr, bias, agg, h if act == L.ACT_RELU else None)
        return h

    @staticmethod
    def backward(ctx, g):
        x, ks, kn, w_csr, bias, agg, h = ctx.saved_tensors
        plan, na, need = ctx.plan, ctx.na, ctx.needs_input_grad
        want_b = bias is not None and need[6]
        # layer 0 (x carries no gradient): both weight-gradient reductions apply the ReLU mask themselves (see _AggregateProject)
        gated = ctx.act == L.ACT_RELU and not need[2] and agg is not None and GATED_AGGREGATE_PROJECT
        g = g.contiguous() if (gated or ctx.act != L.ACT_RELU) else relu_backward(g, h)
        gs, gn = g[:, :na], g[:, na:]
        gx, gks, gba = _linear_grads(x, ks, gs, need[2], need[3], want_b, gate=h[:, :na] if gated else None)
        gkn = gbb = None
        if need[4] or want_b:
            if agg is not None:
                gkn, gbb = gemm_tn(agg, gn, want_bias=want_b, gate=h[:, na:] if gated else None)
                if not need[4]:
                    gkn = None
            else:
                gbb = column_sums(gn)
        if need[2]:
            g_agg = gemm_bias_act(gn, transpose(kn.detach()))
            gx2, _, _ = _aggregate_backward(plan, ctx.mean, x, w_csr, None, g_agg, True, False, False)
            gx = gx + gx2
        gb = torch.cat([gba, gbb]) if want_b else None
        return None, None, gx, gks, gkn, None, gb, None, None


def sage_wide(plan, op, x, ks, kn, w_csr=None, bias=None, act=L.ACT_NONE, rows=None):
    """Differentiable mean / sum GraphSAGE layer body (concat form, aggregation first) with the neighbour half on the fused
    launch, or None when it does not take the shape.  Edge weights are constants here (trainable ones take the un-fused route).
    rows: x's static layout (plan.static_rows) or None / x itself."""
    rows = rows if isinstance(rows, SplitRows) else None
    if op not in (L.SUM, L.MEAN) or not aggregate_gemm_applies(x if rows is None else rows, kn, op):
        return None
    return _SageWide.apply(plan, op == L.MEAN, x, L.as_f32(ks), L.as_f32(kn), w_csr,
                           None if bias is None else L.as_f32(bias), act, rows)


# Gradient of max aggregation (tf.math.unsorted_segment_max, ties share evenly):
#   "mask"  (default) per-edge winner bit masks from the saved arg positions, one gather per edge, bit-reproducible
#   "push"  N*F float atomics from the saved arg positions (summation order = arrival order)
#   "pull"  two row gathers per edge over the transposed plan, bit-reproducible, needs nothing saved (hub graphs)
MAX_GRADIENT_MODE = "mask"
DETERMINISTIC_MAX_GRADIENT = False     # legacy switch: True forces "pull"


def _max_mode():
    return "pull" if DETERMINISTIC_MAX_GRADIENT else MAX_GRADIENT_MODE


class _AggregateMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x, w_csr, passes=None):
        """`passes` (the sharded path, dist/sharded.py::tracked_max_passes): callable(x2, w, out, packed) that runs the
        tracked forward as several launches over consecutive sub-spans of every row (merged in the kernel epilogue) —
        same (out, packed) as the single launch below, so the backward does not know the difference."""
        lib = L.require_gpu()
        x2, ldx = L.row_major_2d(x.detach())
        F = int(x2.shape[1])
        argpos = None
        wd = None if w_csr is None else w_csr.detach()
        packed = None
        if passes is not None:
            out = torch.empty((plan.n_dst, F), dtype=torch.float32, device=x2.device)
            packed = torch.empty((plan.n_dst, F), dtype=torch.int32, device=x2.device)
            passes(x2, wd, out, packed)
            count = None
            ctx.halo_first = getattr(passes, "halo_first", None)     # (n_own, [(lo, hi) per round], callable(j, gx)): see backward
        elif _max_mode() == "mask" and can_track(plan, x2, ldx):
            # the tuned forward walk (8 gathered rows in flight per lane group) with the tie count and the position of the
            # first maximal edge tracked online and written PACKED (one uint32 per element instead of a float count array
            # and an int32 position array): what the mask-form backward reads
            out = torch.empty((plan.n_dst, F), dtype=torch.float32, device=x2.device)
            packed = torch.empty((plan.n_dst, F), dtype=torch.int32, device=x2.device)
            segment_reduce(plan, x2, L.MAX, w_csr=wd, out=out, track=packed)
            count = None
        elif plan.hub_info() is None:
            # one pass: row maxima AND how many edges attain each (the tie count TF's gradient divides by)
            out = torch.empty((plan.n_dst, F), dtype=torch.float32, device=x2.device)
            count = torch.empty_like(out)
            push_ok = _max_mode() in ("mask", "push") and F % 4 == 0 and ldx % 4 == 0 and x2.data_ptr() % 16 == 0
            if push_ok:       # ... and WHICH edge attains it first: the backward becomes N*F atomics instead of E*F gathers
                argpos = torch.empty((plan.n_dst, F), dtype=torch.int32, device=x2.device)
                L.check(lib.tfgx_segment_max_with_arg_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(wd), plan.n_dst,
                                                          L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(count), F,
                                                          L.ptr(argpos), F, L.stream_ptr()),
                        "tfgx_segment_max_with_arg_f32")
            else:
                L.check(lib.tfgx_segment_max_with_count_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(wd), plan.n_dst,
                                                            L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(count), F,
                                                            L.stream_ptr()), "tfgx_segment_max_with_count_f32")
        else:   # skewed graph: the chunked forward keeps long rows off one lane group; count in the backward
            out = segment_reduce(plan, x2, L.MAX, w_csr=wd)
            count = None
        ctx.plan, ctx.count, ctx.argpos, ctx.mode, ctx.packed = plan, count, argpos, _max_mode(), packed
        if passes is None:
            ctx.halo_first = None
        ctx.save_for_backward(x, w_csr, out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.require_gpu()
        plan = ctx.plan
        x, w_csr, out = ctx.saved_tensors
        x2, ldx = L.row_major_2d(x.detach())
        g2, ldg = L.row_major_2d(g.contiguous())
        F = int(x2.shape[1])
        count = ctx.count
        packed = ctx.packed

        def unpack_count():
            return ((packed.to(torch.int64) & 0xFFFFFFFF) >> 16).to(torch.float32)
        if count is None and packed is None:          # skewed graph: hub rows counted chunk-wise
            count = torch.empty_like(out)
            hub_c, nc_c = L.hub_lists(plan)
            sc_c = torch.empty(max(nc_c * F, 1), dtype=torch.float32, device=x2.device) if hub_c is not None else None
            L.check(lib.tfgx_segment_max_count_hub_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w_csr), plan.n_dst,
                                                       L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(count), F,
                                                       None if hub_c is None else ctypes.byref(hub_c), L.ptr(sc_c),
                                                       L.stream_ptr()), "tfgx_segment_max_count_hub_f32")
        gx = None
        if ctx.needs_input_grad[1]:
            gx = torch.empty((int(x2.shape[0]), F), dtype=torch.float32, device=x2.device)   # dense: ld = F below
            aligned = (ctx.argpos is not None or packed is not None) and ldg % 4 == 0 and g2.data_ptr() % 16 == 0
            pt_hub, nc_t = L.hub_lists(_transposed(plan)[0])
            if pt_hub is not None:
                aligned = False        # hub SOURCES: the per-source walks of the mask / push forms would serialise; pull, chunked
            if packed is not None and not aligned:
                count = unpack_count()             # rare fall-back to the pull form: it wants the float count array
            if aligned and packed is not None:
                pt, t2d = _transposed(plan)
                w_t = _transposed_weights(plan, w_csr, t2d)
                ws_bytes = lib.tfgx_segment_max_backward_mask_workspace_bytes(plan.n_dst, plan.num_edges, F)
                ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x2.device)
                n_table = int(x2.shape[0])

                def run(phases, lo, hi):       # masks built / applied to source rows [lo, hi) of the table
                    L.check(lib.tfgx_segment_max_backward_mask_phases_f32(
                        L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(None if w_csr is None else w_csr.detach()), plan.n_dst,
                        plan.num_edges, L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(g2), ldg, None, F, L.ptr(packed), F,
                        pt.row_ptr.data_ptr() + 4 * lo, L.ptr(pt.col), L.ptr(w_t), L.ptr(t2d), hi - lo,
                        gx.data_ptr() + 4 * F * lo, F, L.ptr(ws), ws_bytes, phases, L.stream_ptr()),
                        "tfgx_segment_max_backward_mask_phases_f32 (packed)")
                hf = getattr(ctx, "halo_first", None)
                if hf is not None and 0 < hf[0] < n_table:
                    # sharded table [own | halo]: the halo rows' gradients belong to peers — compute them first, window by
                    # window (one per exchange round), and let each window travel (reverse exchange on the communication
                    # stream) while the following windows and the own rows' part run
                    run(1, 0, 0)                                   # masks + g / count, once
                    for j, (lo, hi) in enumerate(hf[1]):
                        if hi > lo:
                            run(2, lo, hi)
                        hf[2](j, gx)
                    run(2, 0, hf[0])
                else:
                    run(3, 0, n_table)
            elif aligned and ctx.mode == "mask":
                pt, t2d = _transposed(plan)
                w_t = _transposed_weights(plan, w_csr, t2d)
                ws_bytes = lib.tfgx_segment_max_backward_mask_workspace_bytes(plan.n_dst, plan.num_edges, F)
                ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x2.device)
                L.check(lib.tfgx_segment_max_backward_mask_f32(
                    L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(None if w_csr is None else w_csr.detach()), plan.n_dst,
                    plan.num_edges, L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(g2), ldg, L.ptr(count), F, L.ptr(ctx.argpos), F,
                    L.ptr(pt.row_ptr), L.ptr(pt.col), L.ptr(w_t), L.ptr(t2d), int(x2.shape[0]), L.ptr(gx), F, L.ptr(ws),
                    ws_bytes, L.stream_ptr()), "tfgx_segment_max_backward_mask_f32")
            elif aligned:
                L.check(lib.tfgx_segment_max_backward_push_f32(
                    L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(None if w_csr is None else w_csr.detach()), plan.n_dst,
                    int(x2.shape[0]), L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(g2), ldg, L.ptr(count), F,
                    L.ptr(ctx.argpos), F, L.ptr(gx), F, L.stream_ptr()), "tfgx_segment_max_backward_push_f32")
            else:
                pt, t2d = _transposed(plan)
                w_t = _transposed_weights(plan, w_csr, t2d)
                gn = torch.empty_like(out)          # workspace: g / count per destination row
                sc_t = torch.empty(max(nc_t * F, 1), dtype=torch.float32, device=x2.device) if pt_hub is not None else None
                L.check(lib.tfgx_segment_max_backward_hub_f32(
                    L.ptr(pt.row_ptr), L.ptr(pt.col), L.ptr(w_t), pt.n_dst, L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(g2), ldg,
                    L.ptr(count), F, L.ptr(gx), F, plan.n_dst, L.ptr(gn), None if pt_hub is None else ctypes.byref(pt_hub),
                    L.ptr(sc_t), L.stream_ptr()), "tfgx_segment_max_backward_hub_f32")
        gw = None
        if w_csr is not None and ctx.needs_input_grad[2]:
            if count is None:
                count = unpack_count()
            gn2 = g2 / count.clamp(min=1.0)          # TF splits the gradient evenly among tied maxima
            gw = torch.empty_like(w_csr)
            L.check(lib.tfgx_segment_max_backward_w_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), L.ptr(w_csr.detach()),
                                                        plan.n_dst, L.ptr(x2), ldx, F, L.ptr(out), F, L.ptr(gn2), F,
                                                        L.ptr(gw), L.stream_ptr()), "tfgx_segment_max_backward_w_f32")
        return None, gx, gw, None


POOL_MLP_MAX_FUSED = _os.environ.get("TFGX_POOL_MLP_MAX_FUSED", "1") != "0"     # developer A/B


def pool_mlp_max_applies(plan, x, kernel):
    """Can the pooling MLP + max of max-pool GraphSAGE run as autograd._PoolMlpMax with the destination-major weight
    gradient (tfgx_pool_mlp_max_wgrad_f32)?  Dense float32 features that carry NO gradient (layer 0 of a model), shapes the
    kernel is instantiated for, and a plan whose rows the tracked forward can take."""
    if not POOL_MLP_MAX_FUSED or not isinstance(x, torch.Tensor) or x.requires_grad or not torch.is_grad_enabled():
        return False
    lib = L.require_gpu()
    F_in, Fp = int(x.shape[1]), int(kernel.shape[1])
    if not lib.tfgx_pool_mlp_max_wgrad_applies(F_in, Fp):
        return False
    x2, ldx = L.row_major_2d(x)
    if ldx % 4 != 0 or x2.data_ptr() % 16 != 0 or plan.n_dst != plan.n_src:
        return False
    from .plan import gather_friendly_ld
    return _max_mode() == "mask" and can_track(plan, _FutureRows(Fp), gather_friendly_ld(Fp))


class _FutureRows(object):
    """What plan.can_track looks at, for the [n, F] hidden rows that do not exist yet: their width and (torch's allocations are
    256-byte aligned) an aligned address."""

    def __init__(self, F):
        self.shape = (0, F)

    def data_ptr(self):
        return 0


class _PoolMlpMax(torch.autograd.Function):
    """red = max over in-edges of relu(x W + b)[col] — the pooling MLP and the max of max-pool GraphSAGE
    (nn/conv/graph_sage.py:260-269) as ONE operator for features that carry no gradient.  Forward: the MFMA GEMM (bias + ReLU
    in its epilogue, rows at the gather-friendly stride) and the tracked max.  Backward: dW, db straight from the destination
    rows (tfgx_pool_mlp_max_wgrad_f32) — the gradient of the hidden rows, the per-edge winner masks, the ReLU-mask pass and the
    reduction x^T dh of the composed form are never computed."""

    @staticmethod
    def forward(ctx, plan, x, kernel, bias):
        n, Fp = int(x.shape[0]), int(kernel.shape[1])
        h = gemm_bias_act(x.detach(), kernel.detach(), bias=None if bias is None else bias.detach(), act=L.ACT_RELU,
                          out=gather_friendly_empty(n, Fp, x.device))
        red = torch.empty((plan.n_dst, Fp), dtype=torch.float32, device=x.device)
        packed = torch.empty((plan.n_dst, Fp), dtype=torch.int32, device=x.device)
        segment_reduce(plan, h, L.MAX, out=red, track=packed)
        ctx.plan = plan
        ctx.save_for_backward(x, h, red, packed, bias)
        return red

    @staticmethod
    def backward(ctx, g):
        lib = L.require_gpu()
        plan = ctx.plan
        x, h, red, packed, bias = ctx.saved_tensors
        x2, ldx = L.row_major_2d(x.detach())
        h2, ldh = L.row_major_2d(h)
        g2, ldg = L.row_major_2d(g.contiguous())
        F_in, Fp = int(x2.shape[1]), int(h2.shape[1])
        dW = torch.empty((F_in, Fp), dtype=torch.float32, device=x2.device)
        db = torch.empty(Fp, dtype=torch.float32, device=x2.device) if bias is not None else None
        # the launch's work items depend on the graph and on (F_in, Fp) alone: built once, kept on the plan
        items = plan.__dict__.setdefault("_pool_wgrad_items", {})
        key = (F_in, Fp)
        if key not in items:
            nb = lib.tfgx_pool_mlp_max_wgrad_plan_bytes(plan.n_dst, plan.num_edges, F_in, Fp)
            buf = torch.empty(max(nb, 1), dtype=torch.uint8, device=x2.device)
            L.check(lib.tfgx_pool_mlp_max_wgrad_plan(L.ptr(plan.row_ptr), plan.n_dst, plan.num_edges, F_in, Fp, L.ptr(buf), nb,
                                                     L.stream_ptr()), "tfgx_pool_mlp_max_wgrad_plan")
            items[key] = buf
        ws_bytes = lib.tfgx_pool_mlp_max_wgrad_workspace_bytes(plan.n_dst, F_in, Fp)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=x2.device)
        L.check(lib.tfgx_pool_mlp_max_wgrad_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), plan.n_dst, plan.num_edges, L.ptr(x2), ldx, F_in,
                                                L.ptr(h2), ldh, L.ptr(red), Fp, L.ptr(packed), Fp, L.ptr(g2), ldg, Fp,
                                                L.ptr(items[key]), L.ptr(dW), Fp, L.ptr(db), L.ptr(ws), ws_bytes, L.stream_ptr()),
                "tfgx_pool_mlp_max_wgrad_f32")
        need = ctx.needs_input_grad
        return None, None, (dW if need[2] else None), (db if (bias is not None and need[3]) else None)


def pool_mlp_max(plan, x, kernel, bias):
    """max over in-edges of relu(x @ kernel + bias)[col]; differentiable w.r.t. kernel and bias only (ask pool_mlp_max_applies)."""
    return _PoolMlpMax.apply(plan, L.as_f32(x), L.as_f32(kernel), None if bias is None else L.as_f32(bias))


def aggregate(plan, x, op, w_csr=None, self_coef=None, rows=None, bias=None, act=L.ACT_NONE, max_passes=None):
    """Differentiable gather-scale-segment-reduce (sum / mean / max) on `plan`; sum / mean take the layer's bias and
    ReLU in the kernel epilogue (bias: a tensor that may require grad)."""
    if op == L.MAX:
        if self_coef is not None:
            raise NotImplementedError("max aggregation with an implicit self-loop is inference-only")
        h = _AggregateMax.apply(plan, x, w_csr, max_passes)
        if bias is not None:
            h = bias_add(h, bias)
        return torch.relu(h) if act == L.ACT_RELU else h
    return _Aggregate.apply(plan, op == L.MEAN, x, w_csr, self_coef, rows if isinstance(rows, SplitRows) else None,
                            bias, act)


def _linear_grads(x, kernel, g, need_x, need_k, need_b, gate=None):
    """(d/dx, d/dkernel, d/dbias) of x @ kernel + bias given g (which may be a column slice of a wider gradient).
    `gate`: the layer's ReLU output when g is still UN-masked — only valid without d/dx (the reduction kernel applies
    the mask in registers; layer 0 of every model, whose input carries no gradient)."""
    assert gate is None or not need_x
    # d/dx = g @ kernel^T: the forward MFMA kernel with the (small) transposed kernel as B
    gx = gemm_bias_act(g, transpose(kernel.detach())) if need_x else None
    # d/dkernel = x^T @ g and d/dbias = column sums of g: ONE reduction over the node dimension (tfgx_gemm_tn_f32)
    gk = gb = None
    if need_k or need_b:
        gk, gb = gemm_tn(x.detach(), g, want_bias=need_b, gate=gate)
        if not need_k:
            gk = None
    return gx, gk, gb


class _DualLinear(torch.autograd.Function):
    """h = act([x @ ka | r @ kb] + bias): GraphSAGE's concat of the self and neighbour projections (graph_sage.py:43-58)
    with both GEMMs writing straight into their halves of the output (bias + activation in the epilogues) — no concat,
    no bias add, no activation pass; the backward slices the (masked) gradient by column views."""

    @staticmethod
    def forward(ctx, x, ka, r, kb, bias, act):
        n, na, nb = int(x.shape[0]), int(ka.shape[1]), int(kb.shape[1])
        h = torch.empty((n, na + nb), dtype=torch.float32, device=x.device)
        bd = None if bias is None else bias.detach().contiguous()
        gemm_bias_act(x.detach(), ka.detach(), bias=None if bd is None else bd[:na], act=act, out=h[:, :na])
        gemm_bias_act(r.detach(), kb.detach(), bias=None if bd is None else bd[na:], act=act, out=h[:, na:])
        ctx.act, ctx.na = act, na
        ctx.save_for_backward(x, ka, r, kb, bias, h if act == L.ACT_RELU else None)
        return h

    @staticmethod
    def backward(ctx, g):
        x, ka, r, kb, bias, h = ctx.saved_tensors
        na = ctx.na
        need = ctx.needs_input_grad
        relu = ctx.act == L.ACT_RELU
        # no d/dx, d/dr wanted (layer 0: the inputs are data): the ReLU mask is applied inside the reduction kernels
        gated = relu and not need[0] and not need[2] and GATED_WEIGHT_GRADIENTS
        g = relu_backward(g, h) if (relu and not gated) else g.contiguous()
        want_b = bias is not None and need[4]
        gx, gka, gba = _linear_grads(x, ka, g[:, :na], need[0], need[1], want_b, gate=h[:, :na] if gated else None)
        gr, gkb, gbb = _linear_grads(r, kb, g[:, na:], need[2], need[3], want_b, gate=h[:, na:] if gated else None)
        gb = torch.cat([gba, gbb]) if want_b else None
        return gx, gka, gr, gkb, gb, None


class _SageNarrow(torch.autograd.Function):
    """h = act([x @ ks | reduce(w * (x @ kn)[col])] + bias): mean / sum GraphSAGE with the neighbour projection run
    BEFORE the (linear) reduction (nn/conv/graph_sage._self_neighbor_sage), both halves written in place — the GEMM's and
    the aggregation's epilogues carry bias + activation, nothing is concatenated."""

    @staticmethod
    def forward(ctx, plan, mean, x, ks, kn, w_csr, bias, act):
        n, na, nb = int(x.shape[0]), int(ks.shape[1]), int(kn.shape[1])
        h = torch.empty((n, na + nb), dtype=torch.float32, device=x.device)
        bd = None if bias is None else bias.detach().contiguous()
        gemm_bias_act(x.detach(), ks.detach(), bias=None if bd is None else bd[:na], act=act, out=h[:, :na])
        z = gemm_bias_act(x.detach(), kn.detach(), out=gather_friendly_empty(n, nb, x.device))
        segment_reduce(plan, z, L.MEAN if mean else L.SUM, w_csr=None if w_csr is None else w_csr.detach(),
                       out=h[:, na:], act=act, bias=None if bd is None else bd[na:].contiguous())
        ctx.plan, ctx.mean, ctx.act, ctx.na = plan, mean, act, na
        ctx.save_for_backward(x, ks, kn, w_csr, bias, h if act == L.ACT_RELU else None)
        return h

    @staticmethod
    def backward(ctx, g):
        """D = [masked g of the self half | d/dz of the neighbour half] is assembled in ONE buffer, so that both kernels'
        gradients come from one reduction over x (x^T @ D = [d/dks | d/dkn]) and d/dx from one GEMM (D @ [ks | kn]^T)
        instead of two of each plus an add."""
        x, ks, kn, w_csr, bias, h = ctx.saved_tensors
        plan, na, need = ctx.plan, ctx.na, ctx.needs_input_grad
        n, nb = int(g.shape[0]), int(kn.shape[1])
        relu = ctx.act == L.ACT_RELU
        want_b = bias is not None and need[6]
        D = torch.empty((n, na + nb), dtype=torch.float32, device=g.device)
        gn = gather_friendly_empty(n, nb, g.device)      # gathered by the transposed pass: line-friendly stride
        if relu:
            relu_backward(g[:, :na], h[:, :na], into=D[:, :na])
            relu_backward(g[:, na:], h[:, na:], into=gn)
        else:
            D[:, :na] = g[:, :na]
            gn.copy_(g[:, na:])
        gbb = column_sums(gn) if want_b else None
        if ctx.mean:
            gn.div_(plan.in_degree().clamp(min=1).to(gn.dtype).unsqueeze(1))
        pt, t2d = _transposed(plan)
        segment_reduce(pt, gn, L.SUM, w_csr=_transposed_weights(plan, w_csr, t2d), out=D[:, na:])
        gks = gkn = gb = gx = None
        if need[3] or need[4] or want_b:
            gW, gb_all = gemm_tn(x.detach(), D, want_bias=want_b)
            gks, gkn = (gW[:, :na] if need[3] else None), (gW[:, na:] if need[4] else None)
            if want_b:
                gb = torch.cat([gb_all[:na], gbb])
        if need[2]:
            gx = gemm_bias_act(D, transpose(torch.cat([ks.detach(), kn.detach()], dim=1)))
        return None, None, gx, gks, gkn, None, gb, None


def sage_narrow(plan, op, x, ks, kn, w_csr=None, bias=None, act=L.ACT_NONE):
    """Fused differentiable mean / sum GraphSAGE layer body (concat form, neighbour projection first).  The edge weights
    are treated as constants (callers route trainable edge weights through the un-fused operators)."""
    return _SageNarrow.apply(plan, op == L.MEAN, x, L.as_f32(ks), L.as_f32(kn), w_csr,
                             None if bias is None else L.as_f32(bias), act)


def dual_linear(x, ka, r, kb, bias=None, act=L.ACT_NONE):
    return _DualLinear.apply(x, L.as_f32(ka), r, L.as_f32(kb), None if bias is None else L.as_f32(bias), act)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, act, gathered=False):
        """gathered=True: the rows are gathered by an aggregation next — written at a line-friendly stride."""
        dst = gather_friendly_empty(int(x.shape[0]), int(kernel.shape[1]), x.device) if gathered else None
        out = gemm_bias_act(x.detach(), kernel.detach(), bias=None if bias is None else bias.detach(), act=act, out=dst)
        ctx.act = act
        ctx.save_for_backward(x, kernel, bias, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, kernel, bias, out = ctx.saved_tensors
        relu = ctx.act == L.ACT_RELU
        gated = relu and not ctx.needs_input_grad[0] and GATED_WEIGHT_GRADIENTS      # no d/dx: mask applied inside the reduction kernel
        g = relu_backward(g, out) if (relu and not gated) else g.contiguous()
        gx, gk, gb = _linear_grads(x, kernel, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                   bias is not None and ctx.needs_input_grad[2], gate=out if gated else None)
        return gx, gk, gb, None, None


def linear(x, kernel, bias=None, act=L.ACT_NONE, gathered=False):
    """act(x @ kernel + bias): forward on the MFMA kernel, differentiable.  gathered=True when an aggregation reads the
    result next (plan.gather_friendly_ld)."""
    return _Linear.apply(x, L.as_f32(kernel), None if bias is None else L.as_f32(bias), act, gathered)


class _ProjectQKV(torch.autograd.Function):
    """Q = act(x Wq + bq), K = act(x Wk + bk), V = x W (nn/conv/gat.py:52-70) as ONE differentiable operator: the forward reads
    x twice ([Q | K] in one GEMM when they share the activation, V in a second one with its gather-friendly stride), the
    backward ONCE — the masked dQ, dK and dV are laid side by side in one [n, 2A + U] matrix D, so that all three weight
    gradients and both bias gradients come from one reduction over x (x^T D) and d/dx from one GEMM (D [Wq | Wk | W]^T).
    Three separate linear layers read the 561 MB Reddit-shaped x six times per training step; this reads it three times."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, act):
        n, A, U = int(x.shape[0]), int(wq.shape[1]), int(wv.shape[1])
        xd = x.detach()
        w_qk = torch.cat([wq.detach(), wk.detach()], dim=1).contiguous()
        b_qk = None if bq is None else torch.cat([bq.detach().reshape(-1), bk.detach().reshape(-1)])
        qk = gemm_bias_act(xd, w_qk, bias=b_qk, act=act)
        V = gemm_bias_act(xd, wv.detach(), out=gather_friendly_empty(n, U, x.device))
        # K rows narrower than a 128-byte line are gathered per edge: inside [Q | K] a line holds half as many of them
        # (nn/conv/gat._project_qkv); a copy of [n, A] floats costs ~10 us
        K = qk[:, A:].contiguous() if A <= 16 else qk[:, A:]
        ctx.act, ctx.A = act, A
        ctx.save_for_backward(x, wq, wk, wv, bq, qk if act == L.ACT_RELU else None)
        return qk[:, :A], K, V

    @staticmethod
    def backward(ctx, gQ, gK, gV):
        x, wq, wk, wv, bq, qk = ctx.saved_tensors
        A, U, n = ctx.A, int(wv.shape[1]), int(x.shape[0])
        need = ctx.needs_input_grad
        D = torch.empty((n, 2 * A + U), dtype=torch.float32, device=x.device)
        if ctx.act == L.ACT_RELU:
            relu_backward(gQ, qk[:, :A], into=D[:, :A])
            relu_backward(gK, qk[:, A:], into=D[:, A:2 * A])
        else:
            D[:, :A] = gQ
            D[:, A:2 * A] = gK
        D[:, 2 * A:] = gV
        gwq = gwk = gwv = gbq = gbk = gx = None
        if need[1] or need[2] or need[3] or need[4] or need[5]:
            gW, gb = gemm_tn(x.detach(), D, want_bias=bq is not None and (need[2] or need[4]))
            gwq, gwk, gwv = gW[:, :A], gW[:, A:2 * A], gW[:, 2 * A:]
            if gb is not None:
                gbq, gbk = gb[:A], gb[A:2 * A]
        if need[0]:
            gx = gemm_bias_act(D, transpose(torch.cat([wq.detach(), wk.detach(), wv.detach()], dim=1)))
        return gx, gwq, gbq, gwk, gbk, gwv, None


def project_qkv(x, wq, bq, wk, bk, wv, act):
    """Differentiable (Q, K, V) of a GAT layer on dense features; Q and K share the (fused-code) activation `act`."""
    f = L.as_f32
    return _ProjectQKV.apply(x, f(wq), None if bq is None else f(bq), f(wk), None if bk is None else f(bk), f(wv), act)


class _GatAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, num_heads, Q, K, V, drop_rate, drop_seed, scale_d=None, passes=None):
        """`passes` (the sharded path, dist/sharded.py::_gat_attention_spans): callable(Q, K, V, stats) -> out that runs the
        forward as span passes + merge under a halo exchange; same (out, stats), so the backward is unchanged."""
        from .nn.conv.gat import gat_attention, query_sums_apply, SOURCE_BLOCK_STATS
        stats = torch.empty((plan.n_dst, 2 * num_heads), dtype=torch.float32, device=V.device)
        ctx.halo_first = getattr(passes, "halo_first", None)   # (n_own, [(lo, hi) per round], callable(j, d[K | V])): see backward
        qsums = None
        if passes is not None:
            assert float(drop_rate) == 0.0 and scale_d is None
            out = passes(Q.detach(), K.detach(), V.detach(), stats)
        else:
            # one attention unit per head (the demo's literal layer): the walk also accumulates the two sums dQ is a per-row
            # expression of (tfgx_gat_args.qgrad_t) — the backward then runs no destination pass
            if ctx.needs_input_grad[2] and query_sums_apply(plan, Q, V, num_heads, float(drop_rate)):
                qsums = (torch.empty((plan.n_dst, int(V.shape[1])), dtype=torch.float32, device=V.device),
                         torch.empty((plan.n_dst, num_heads), dtype=torch.float32, device=V.device))
                SOURCE_BLOCK_STATS["query_sum_forwards"] = SOURCE_BLOCK_STATS.get("query_sum_forwards", 0) + 1
            out = gat_attention(plan, Q.detach(), K.detach(), V.detach(), num_heads, True, stats_ml=stats,
                                drop_rate=drop_rate, drop_seed=drop_seed, scale_d=scale_d, qsums=qsums)
        ctx.plan, ctx.H, ctx.drop = plan, num_heads, (float(drop_rate), drop_seed)      # (seed: host int or device tensor)
        ctx.scale_d = scale_d
        ctx.has_qsums = qsums is not None
        ctx.save_for_backward(Q, K, V, out, stats, *(qsums or ()))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.require_gpu()
        plan, H = ctx.plan, ctx.H
        Q, K, V, out, stats = ctx.saved_tensors[:5]
        qsums = ctx.saved_tensors[5:7] if ctx.has_qsums else None
        Q2, ldq = L.row_major_2d(Q.detach())
        K2, ldk = L.row_major_2d(K.detach())
        V2, ldv = L.row_major_2d(V.detach())
        g2, ldg = L.row_major_2d(g.contiguous())
        n, A, W = plan.n_dst, int(Q2.shape[1]), int(V2.shape[1])
        # one sweep over the destination rows: D = <dO, O> per head AND the packed rows the source pass gathers — per
        # edge it needs the DESTINATION's dO, Q, (m, l) and D; interleaved into one row per destination, padded to whole
        # 128-byte lines, that is one burst of P*4 bytes per edge instead of four gathers (H=8, A=8, U=64: 96 floats =
        # 3 lines instead of 5).  Round 6: behind dO the row holds ONE block per head, [Q | m | 1 / (l + 1e-8) | D | pad] in
        # roundup4(d + 3) floats — every lane of a head then fetches its scalars with 16-byte loads of the same bytes: three
        # line REQUESTS per edge instead of five (tfgx_gat_backward_args.head_pack)
        HB = -(-(A // H + 3) // 4) * 4
        W4 = -(-W // 4) * 4                                      # the head blocks start on a 16-byte boundary of the row
        P = -(-(W4 + H * HB) // 32) * 32
        pack = torch.empty((n, P), dtype=torch.float32, device=g2.device)
        dsum = torch.empty((n, H), dtype=torch.float32, device=g2.device)
        out2, ldo = L.row_major_2d(out)
        L.check(lib.tfgx_gat_pack_dst_heads_f32(L.ptr(g2), ldg, L.ptr(out2), ldo, L.ptr(Q2), ldq, L.ptr(stats), n, H, A // H,
                                                W // H, L.ptr(pack), P, L.ptr(dsum), L.stream_ptr()),
                "tfgx_gat_pack_dst_heads_f32")
        pt, t2d = _transposed(plan)
        # dense outputs (K2 / V2 may be column slices of a wider table — the sharded [K | V] halo table)
        dev = g2.device
        gq = torch.empty((n, A), dtype=torch.float32, device=dev)
        hf = getattr(ctx, "halo_first", None)
        n_tab = int(K2.shape[0])
        if hf is not None and int(V2.shape[0]) == n_tab:
            # sharded [K | V] table: d[K | V] in ONE buffer (its halo rows travel back as one exchange), gk / gv are its column
            # blocks — what autograd's slice backward would assemble anyway
            gkv = torch.empty((n_tab, A + W), dtype=torch.float32, device=dev)
            gk, gv = gkv[:, :A], gkv[:, A:]
        else:
            hf, gkv = None, None
            gk = torch.empty((n_tab, A), dtype=torch.float32, device=dev)
            gv = torch.empty((int(V2.shape[0]), W), dtype=torch.float32, device=dev)
        a = L.GatBackwardArgs()
        a.row_ptr, a.col, a.n_dst = plan.row_ptr.data_ptr(), plan.col.data_ptr(), n
        a.row_ptr_t, a.dst_t, a.n_src = pt.row_ptr.data_ptr(), pt.col.data_ptr(), pt.n_dst
        a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = Q2.data_ptr(), ldq, K2.data_ptr(), ldk, V2.data_ptr(), ldv
        a.grad_out, a.ld_grad_out = g2.data_ptr(), ldg
        a.stats_ml, a.dsum = stats.data_ptr(), dsum.data_ptr()
        a.H, a.d, a.dv, a.add_self_loop = H, A // H, W // H, 1
        a.scale = math.sqrt(float(A // H if ctx.scale_d is None else ctx.scale_d))
        a.grad_q, a.ld_grad_q = gq.data_ptr(), A
        a.grad_k, a.ld_grad_k = gk.data_ptr(), int(gk.stride(0))
        a.grad_v, a.ld_grad_v = gv.data_ptr(), int(gv.stride(0))
        ro, ro_t = plan.row_order(), pt.row_order()        # skewed graphs: degree-ordered walks (results unchanged)
        a.row_order = 0 if ro is None else ro.data_ptr()
        a.row_order_t = 0 if ro_t is None else ro_t.data_ptr()
        if ctx.drop[0] > 0.0:      # regenerate the forward's keep mask: same seed, forward-CSR edge positions
            from .nn.conv.gat import _set_drop
            _set_drop(a, ctx.drop[0], ctx.drop[1], plan.num_edges)
            a.edge_pos_t = t2d.data_ptr()
        # power-law graphs: rows too long for one lane group are walked chunk-wise (the plans' hub lists) and their chunk
        # partials added in order — dQ over the forward plan's hub destinations, dK / dV over the transposed plan's hub sources
        hub_d, nc_d = L.hub_lists(plan)
        sc_d = torch.empty(max(nc_d * A, 1), dtype=torch.float32, device=dev) if hub_d is not None else None
        # dense graphs: both passes in chained launches over the blocks of the OTHER endpoint (nn/conv/gat.source_block_count:
        # each launch gathers rows of one block, served by the L2 of every XCD; gradients accumulate block by block)
        from .nn.conv.gat import source_block_count, SOURCE_BLOCK_STATS
        from .nn.conv import gat as G_
        d_h, dv_h = A // H, W // H
        blocks_ok = (ctx.drop[0] <= 0.0 and hf is None and hub_d is None and ro is None and ro_t is None and
                     d_h in (1, 2, 4, 8, 16, 32) and dv_h % 4 == 0 and dv_h // 4 <= 64 and
                     (((dv_h // 4) & (dv_h // 4 - 1)) == 0 or H == 1) and ldv % 4 == 0 and ldg % 4 == 0 and
                     V2.data_ptr() % 16 == 0 and g2.data_ptr() % 16 == 0)
        kb_d = source_block_count(plan, A, W) if blocks_ok else 1
        blk_d = plan.source_blocks(kb_d) if kb_d >= 2 else None
        if qsums is not None:
            # dQ from the forward's sums: (<dO, T> - D S) / scale per (row, head) — no walk over the edges
            L.check(lib.tfgx_gat_query_grad_d1_f32(L.ptr(g2), ldg, L.ptr(qsums[0]), W, L.ptr(qsums[1]), L.ptr(dsum), n, H, W // H,
                                                   a.scale, L.ptr(gq), A, L.stream_ptr()), "tfgx_gat_query_grad_d1_f32")
            SOURCE_BLOCK_STATS["query_sum_backwards"] = SOURCE_BLOCK_STATS.get("query_sum_backwards", 0) + 1
        elif blk_d is not None:
            rpk, col_k = blk_d
            for b in range(kb_d):
                a.col = col_k.data_ptr()
                a.span_begin, a.span_end, a.span_stride = rpk[b:].data_ptr(), rpk[b + 1:].data_ptr(), kb_d
                a.accumulate, a.add_self_loop = (1 if b else 0), (1 if b == kb_d - 1 else 0)
                L.check(lib.tfgx_gat_backward_dst_hub_f32(ctypes.byref(a), None, None, L.stream_ptr()),
                        "tfgx_gat_backward_dst_hub_f32 (source block)")
            a.col, a.span_begin, a.span_end, a.span_stride, a.accumulate, a.add_self_loop = plan.col.data_ptr(), 0, 0, 0, 0, 1
            SOURCE_BLOCK_STATS["backward_launches"] = SOURCE_BLOCK_STATS.get("backward_launches", 0) + kb_d
        else:
            L.check(lib.tfgx_gat_backward_dst_hub_f32(ctypes.byref(a), None if hub_d is None else ctypes.byref(hub_d),
                                                      L.ptr(sc_d), L.stream_ptr()), "tfgx_gat_backward_dst_hub_f32")
        a.grad_out, a.ld_grad_out = pack.data_ptr(), P            # dO out of the packed rows; Q / (m, l) / D stay the dense arrays
        a.head_pack, a.ld_head_pack = pack.data_ptr() + 4 * W4, P  # (read by the one-lane kernels of odd head geometries only)
        hub_s, nc_s = L.hub_lists(pt)
        sc_s = torch.empty(max(nc_s * (A + W), 1), dtype=torch.float32, device=dev) if hub_s is not None else None
        if hf is not None and hub_s is None and 0 < hf[0] < n_tab:
            # halo rows of the table first (sources that are no destinations: no self-loop, window-relative row indices:
            # the K / V / gradient pointers move with the window), their gradients start travelling, then the own rows
            n_own = int(hf[0])
            full = (a.row_ptr_t, a.n_src, a.k, a.v, a.grad_k, a.grad_v, a.add_self_loop, a.row_order_t)
            a.add_self_loop, a.row_order_t = 0, 0
            for j, (lo, hi) in enumerate(hf[1]):               # one window per exchange round (the halo rows are round-major)
                if hi > lo:
                    a.row_ptr_t = pt.row_ptr.data_ptr() + 4 * lo
                    a.n_src = hi - lo
                    a.k, a.v = K2.data_ptr() + 4 * ldk * lo, V2.data_ptr() + 4 * ldv * lo
                    a.grad_k = gk.data_ptr() + 4 * int(gk.stride(0)) * lo
                    a.grad_v = gv.data_ptr() + 4 * int(gv.stride(0)) * lo
                    L.check(lib.tfgx_gat_backward_src_hub_f32(ctypes.byref(a), None, None, L.stream_ptr()),
                            "tfgx_gat_backward_src_hub_f32 (halo rows)")
                hf[2](j, gkv)
            a.row_ptr_t, a.n_src, a.k, a.v, a.grad_k, a.grad_v, a.add_self_loop, a.row_order_t = full
            a.n_src, a.row_order_t = n_own, 0
            L.check(lib.tfgx_gat_backward_src_hub_f32(ctypes.byref(a), None, None, L.stream_ptr()),
                    "tfgx_gat_backward_src_hub_f32 (own rows)")
        else:
            # the source pass gathers the packed DESTINATION rows (P floats each): blocks of destinations
            kb_s = source_block_count(pt, P, 0) if (blocks_ok and hub_s is None and gv.stride(0) % 4 == 0) else 1
            if kb_s >= 2 and G_.DESTINATION_BLOCKS is not None:
                kb_s = max(int(G_.DESTINATION_BLOCKS), 1)       # developer A/B (tools/r06/sweep_gat_bwd_blocks.py)
            blk_s = pt.source_blocks(kb_s) if kb_s >= 2 else None
            if blk_s is not None:
                rpk_t, dst_k = blk_s
                for b in range(kb_s):
                    a.dst_t = dst_k.data_ptr()
                    a.span_begin, a.span_end, a.span_stride = rpk_t[b:].data_ptr(), rpk_t[b + 1:].data_ptr(), kb_s
                    a.accumulate, a.add_self_loop = (1 if b else 0), (1 if b == kb_s - 1 else 0)
                    L.check(lib.tfgx_gat_backward_src_hub_f32(ctypes.byref(a), None, None, L.stream_ptr()),
                            "tfgx_gat_backward_src_hub_f32 (destination block)")
                SOURCE_BLOCK_STATS["backward_launches"] = SOURCE_BLOCK_STATS.get("backward_launches", 0) + kb_s
            else:
                L.check(lib.tfgx_gat_backward_src_hub_f32(ctypes.byref(a), None if hub_s is None else ctypes.byref(hub_s),
                                                          L.ptr(sc_s), L.stream_ptr()), "tfgx_gat_backward_src_hub_f32")
        return None, None, gq, gk, gv, None, None, None, None


def gat_attention(plan, Q, K, V, num_heads, drop_rate=0.0, drop_seed=0, scale_d=None, passes=None):
    """Differentiable fused attention (self-loop edge appended, as nn/conv/gat.py:43); drop_rate > 0 = training-time
    dropout of the attention weights (gat.py:85).  scale_d: the per-head width the scores are scaled by when Q / K
    arrive zero-padded per head (nn/conv/gat._kernel_widths)."""
    return _GatAttention.apply(plan, num_heads, Q, K, V, float(drop_rate),
                               drop_seed if isinstance(drop_seed, torch.Tensor) else int(drop_seed), scale_d, passes)


def edge_attr_csr(plan, edge_attr, cache=None):
    """edge attribute (caller's edge order) -> the plan's CSR order; a differentiable permutation when the attribute is
    being tracked, the memoised kernel copy (plan.edge_weight_csr) otherwise."""
    from .plan import edge_weight_csr
    if edge_attr is None:
        return None
    if needs_grad(edge_attr):
        return L.as_f32(edge_attr)[plan.perm.long()]
    return edge_weight_csr(plan, edge_attr, cache)


def gather(x, idx):
    """x[idx] (tf.gather): the gather kernel, or torch indexing when x is being tracked."""
    from .plan import gather_rows
    if needs_grad(x):
        return L.as_f32(x)[L.as_i32(idx).long()]
    return gather_rows(x, idx)


class _SegmentSoftmax(torch.autograd.Function):
    """out = exp(d - stop_gradient(segmax)) / (segsum + 1e-8)  (nn/kernel/segment.py:26-33):
    d out_i / d d_j = [i == j] out_i - out_i out_j inside a segment, so grad = out * (g - segsum(out * g))."""

    @staticmethod
    def forward(ctx, plan, ids, d):
        lib = L.require_gpu()
        dd = d.detach().contiguous()
        out = torch.empty_like(dd)
        H = 1 if dd.dim() == 1 else int(dd.shape[1])
        if int(ids.shape[0]):
            L.edge_softmax(plan, dd, H, out)
        ctx.plan, ctx.ids = plan, ids
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        s = out * g
        s2 = s if s.dim() == 2 else s.unsqueeze(1)
        # plan.col is all zeros for this plan: reduce the E explicit rows through perm instead
        seg = segment_reduce(ctx.plan, s2.contiguous(), L.SUM, col=ctx.plan.perm)
        back = seg[ctx.ids.long()]
        return None, None, s - out * (back if s.dim() == 2 else back[:, 0])


def segment_softmax(plan, ids, data):
    return _SegmentSoftmax.apply(plan, ids, data)


class _BiasAdd(torch.autograd.Function):
    """h + bias (broadcast over rows).  torch's own add would do — but its backward reduces the [N, units] gradient with
    g.sum(0), which runs odd widths (47 classes of ogbn-products, 41 of Reddit) at 24 GB/s: 19 ms per step at products shape.
    The backward here is the two-phase column-sum kernel (plan.column_sums)."""

    @staticmethod
    def forward(ctx, h, bias):
        return h + bias

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.needs_input_grad[0] else None), (column_sums(g) if ctx.needs_input_grad[1] else None)


def bias_add(h, bias):
    """h + bias for a [N, units] activation and a [units] bias (differentiable; None bias: h itself)."""
    if bias is None:
        return h
    b = L.as_f32(bias)
    if h.dim() == 2 and b.dim() == 1 and h.is_cuda and needs_grad(b):
        return _BiasAdd.apply(h, b)
    return h + b


def apply_activation(h, act, post):
    if act == L.ACT_RELU:
        h = torch.relu(h)
    return post(h) if post is not None else h
