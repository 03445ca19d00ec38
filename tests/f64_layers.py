# coding=utf-8
"""TEST INFRASTRUCTURE ONLY — float64 restatements of the reference's LAYERS as plain torch ops, differentiated by torch
autograd on the GPU, sized for BASELINE.json's full shapes (114 M / 123 M edges).

The reference trains with tf.GradientTape over its forward composition (demo/demo_gat.py:66-75, demo/demo_graph_sage.py);
the role of "the true gradient" is played here by torch's float64 autograd over the same composition — gather, multiply,
sorted-segment sum / max, exp, division, matmul, relu — written from the reference's lines (cited per function) with nothing
of tf_geometric_amd in it.  Edge-sized intermediates are [E, chunk] float64, so every function walks the feature columns (or
the heads) in chunks and frees each chunk's autograd graph before the next: the peak stays at a few tens of GB of the 288.
Segments are reduced with torch.segment_reduce over destination-SORTED edges (no atomics: a power-law hub row of 800 k
in-edges costs 0.1 s instead of 16 s of colliding float64 atomics; ties of a max share its gradient evenly, as
math_grad._UnsortedSegmentMinOrMaxGrad does).

Where the derivative does not exist at float32 resolution.  ReLU and max are piecewise linear: an output's pre-activation
within float32 rounding of 0, or a row maximum whose runner-up is within float32 rounding of it, has a derivative that depends
on which side the ROUNDING fell — a float32 implementation (TensorFlow's included) and a float64 one legitimately disagree
there, and one flipped decision moves a whole gradient row by O(1).  Among 6e8 outputs there are always a few dozen such
elements, so each function (i) decides from its own float64 forward which upstream-gradient entries would flow through such a
spot (margins: arguments `*_margin`, 20-100 x the float32 forward error) and ZEROES them — it returns the gradient it used,
`G_eff`, and the product is differentiated against the same `G_eff`; (ii) for inner ReLUs that no upstream entry can isolate
(GAT's Q / K projections) the INPUT rows whose pre-activations sit on the kink are redrawn first (`redraw_kink_rows`).
Both are properties of the float64 reference alone: nothing of the product's output is consulted.

Used by tests/test_gpu_fullsize.py only (the small-shape gradient tests keep their own CPU float64 references).
"""
import math

import torch

F32_LOWEST = -3.4028234663852886e38          # tf.math.unsorted_segment_max of an empty segment (float32 lowest)


def _leaf(a, dev, grad=True):
    t = torch.as_tensor(a).to(device=dev, dtype=torch.float64).clone()
    return t.requires_grad_(grad)


class SortedEdges(object):
    """The edge list sorted (stably) by destination: segment k = the in-edges of node k."""

    def __init__(self, ei, n, self_loops=False):
        row, col = ei[0].long(), ei[1].long()
        if self_loops:                          # nn/conv/gat.py:43 / utils/graph_utils.py:350-366: (i, i) appended AFTER the edges
            ar = torch.arange(n, device=row.device)
            row, col = torch.cat([row, ar]), torch.cat([col, ar])
        self.order = torch.sort(row, stable=True).indices
        self.row, self.col = row[self.order], col[self.order]
        self.lengths = torch.bincount(row, minlength=n)
        self.n = n

    def edge_attr(self, a):
        return a[self.order]

    def seg_sum(self, data):
        return torch.segment_reduce(data, "sum", lengths=self.lengths, axis=0, unsafe=True)

    def seg_max(self, data, initial):
        """Value only.  (torch's own backward of this op is NOT used: at 123 M x 16 elements it hands a tied maximum's gradient
        to the tied entries more than once — tools/r06/diag_max_backward.py, hand-checked element; see seg_max_tf.)"""
        return torch.segment_reduce(data.detach(), "max", lengths=self.lengths, axis=0, unsafe=True, initial=initial)

    def seg_max_tf(self, data, initial):
        """Differentiable segment max with TensorFlow's registered gradient, written out (math_grad._UnsortedSegmentMinOrMaxGrad)."""
        return _SegmentMaxTF.apply(data, self, initial)


class _SegmentMaxTF(torch.autograd.Function):
    """tf.math.unsorted_segment_max and its registered gradient: is_selected = (data == gather(out, ids)); num_selected =
    segment_sum(is_selected); the segment's gradient divided by num_selected goes to every selected entry."""

    @staticmethod
    def forward(ctx, data, g, initial):
        out = g.seg_max(data, initial)
        ctx.g = g
        ctx.save_for_backward(data, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        data, out = ctx.saved_tensors
        g = ctx.g
        sel = data == out[g.row]
        num = g.seg_sum(sel.to(grad.dtype))
        return sel * (grad / num.clamp(min=1.0))[g.row], None, None


def segment_sum_columns(g, x, w, chunk=25):
    """sum over edges e of w_e * x[col_e] into row_e (nn/kernel/map_reduce.py:15-16 with gcn_mapper, nn/conv/gcn.py:221-222),
    differentiable wrt x; [E, chunk] float64 alive at a time.  g: SortedEdges, w in g's order."""
    outs = []
    for c0 in range(0, int(x.shape[1]), chunk):
        src = x[:, c0:c0 + chunk][g.col]
        if w is not None:
            src = src * w[:, None]
        outs.append(g.seg_sum(src))
        del src
    return torch.cat(outs, dim=1)


def _outer_relu(pre, G, margin):
    """out = relu(pre); the upstream gradient with the entries at the kink (|pre| <= margin) zeroed; the gradient entering pre."""
    G_eff = G.double() * (pre.detach().abs() > margin)
    return torch.relu(pre), G_eff


def gcn_layer(x32, ei, w32, kernel, bias, G, x_grad=True, kink_margin=1e-4):
    """relu(A_hat (x W) + b), A_hat = D^-1/2 (A + I) D^-1/2 (nn/conv/gcn.py:32-130 defaults: renorm, sym, add_self_loop;
    :225-290).  -> (out, gradients of <out, G_eff>, G_eff float32)."""
    dev = x32.device
    n = int(x32.shape[0])
    g = SortedEdges(ei, n)
    x = _leaf(x32, dev, x_grad)
    W, b = _leaf(kernel, dev), _leaf(bias, dev)
    w = g.edge_attr(w32.double())
    deg = g.seg_sum(w) + 1.0                                                                 # rows of A + I  (gcn.py:77-80)
    dis = deg.pow(-0.5)
    dis[~torch.isfinite(dis)] = 0.0
    what = dis[g.row] * w * dis[g.col]                                                       # gcn.py:85-91
    # (A_hat x) W == A_hat (x W) in exact arithmetic; the reference multiplies first (gcn.py:272-280) — float64 either way
    agg = segment_sum_columns(g, x, what) + (dis * dis)[:, None] * x
    out, G_eff = _outer_relu(agg @ W + b, G, kink_margin)
    out.backward(G_eff)
    return out.detach(), {"x": x.grad, "kernel": W.grad, "bias": b.grad}, G_eff.float()


def mean_sage_layer(x32, ei, w32, self_kernel, neighbor_kernel, bias, G, x_grad=True, kink_margin=1e-4):
    """relu([x W_self | mean_e(w_e x[col_e]) W_neigh] + b) (nn/conv/graph_sage.py:9-60, concat=True)."""
    dev = x32.device
    n = int(x32.shape[0])
    g = SortedEdges(ei, n)
    x = _leaf(x32, dev, x_grad)
    Ws, Wn, b = _leaf(self_kernel, dev), _leaf(neighbor_kernel, dev), _leaf(bias, dev)
    cnt = g.lengths.clamp(min=1).double()
    m = segment_sum_columns(g, x, g.edge_attr(w32.double())) / cnt[:, None]
    out, G_eff = _outer_relu(torch.cat([x @ Ws, m @ Wn], dim=1) + b, G, kink_margin)
    out.backward(G_eff)
    return out.detach(), {"x": x.grad, "self_kernel": Ws.grad, "neighbor_kernel": Wn.grad, "bias": b.grad}, G_eff.float()


def max_pool_sage_layer(x32, ei, self_kernel, mlp_kernel, mlp_bias, neighs_kernel, bias, G, x_grad=True, chunk=16,
                        kink_margin=1e-4, tie_margin=2e-5):
    """relu([x W_self | max_e relu(x W_mlp + b_mlp)[col_e] W_neigh] + b) (nn/conv/graph_sage.py:228-287: a provided
    edge_weight is replaced by ones :253-254, the same activation after the MLP and at the end :263,:281-282).
    The 512 pooled columns are walked `chunk` at a time: forward without a graph first (the pooled matrix becomes a leaf of
    the dense tail), then each chunk again with its graph, differentiated and freed.
    Rows whose pooling is decided inside `tie_margin` in some column — the largest MLP pre-activation of the row within the
    margin of 0 (ReLU kink at the winner), or positive with a DIFFERENT value within the margin below it (near-tie; exact ties,
    e.g. a duplicated edge, are the same on both sides and stay) — and rows without in-edges (float32 lowest through the next
    GEMM: overflow, see the forward test) receive no upstream gradient."""
    dev = x32.device
    n = int(x32.shape[0])
    g = SortedEdges(ei, n)
    x = _leaf(x32, dev, x_grad)
    Ws, Wm, bm = _leaf(self_kernel, dev), _leaf(mlp_kernel, dev), _leaf(mlp_bias, dev)
    Wn, b = _leaf(neighs_kernel, dev), _leaf(bias, dev)
    width = int(Wm.shape[1])

    def pooled(c0):
        h = torch.relu(x @ Wm[:, c0:c0 + chunk] + bm[c0:c0 + chunk])
        return g.seg_max_tf(h[g.col], F32_LOWEST)

    ambiguous = g.lengths == 0
    with torch.no_grad():
        reds = []
        for c0 in range(0, width, chunk):
            pre = (x @ Wm[:, c0:c0 + chunk] + bm[c0:c0 + chunk])[g.col]
            top = g.seg_max(pre, -math.inf)
            runner = g.seg_max(torch.where(pre == top[g.row], torch.full_like(pre, -math.inf), pre), -math.inf)
            near = (top.abs() <= tie_margin) | ((top > 0) & (top - runner <= tie_margin))
            ambiguous |= near.any(1)
            reds.append(torch.relu(top))
            del pre, top, runner, near
        red = torch.cat(reds, dim=1)
        red[g.lengths == 0] = F32_LOWEST
    red.requires_grad_(True)
    out, G_eff = _outer_relu(torch.cat([x @ Ws, red @ Wn], dim=1) + b, G, kink_margin)
    G_eff = G_eff * (~ambiguous)[:, None]
    out.backward(G_eff)
    for c0 in range(0, width, chunk):
        r = pooled(c0)
        r.backward(red.grad[:, c0:c0 + chunk])
        del r
    grads = {"x": x.grad, "self_kernel": Ws.grad, "mlp_kernel": Wm.grad, "mlp_bias": bm.grad, "neighs_kernel": Wn.grad,
             "bias": b.grad}
    return out.detach(), grads, G_eff.float(), ambiguous


def redraw_kink_rows(x32, pairs, margin=2e-5, seed=77, max_rounds=8):
    """Rows of x whose pre-activation x @ W + b (for any (W, b) of `pairs`) lies within `margin` of the ReLU kink are drawn
    again (standard normal, as the synthetic features are) until none does.  -> (x32 copy, rows redrawn)."""
    dev = x32.device
    x = x32.clone()
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    redrawn = 0
    for _ in range(max_rounds):
        bad = torch.zeros(int(x.shape[0]), dtype=torch.bool, device=dev)
        for W, b in pairs:
            pre = x.double() @ torch.as_tensor(W).to(dev).double() + torch.as_tensor(b).to(dev).double()
            bad |= (pre.abs() <= margin).any(1)
        k = int(bad.sum())
        if k == 0:
            return x, redrawn
        x[bad] = torch.randn(k, int(x.shape[1]), generator=gen, device=dev)
        redrawn += k
    raise RuntimeError("redraw_kink_rows: still {} rows on a kink after {} rounds".format(k, max_rounds))


def gat_forward(x32, ei, wq, bq, wk, bk, wv, bias, num_heads, dtype=torch.float64):
    """The forward alone in `dtype` (float32: the reference's own formulation evaluated op for op at the reference's width)."""
    return _gat(x32, ei, wq, bq, wk, bk, wv, bias, num_heads, None, False, dtype, 0.0)[0]


def gat_layer(x32, ei, wq, bq, wk, bk, wv, bias, num_heads, G, x_grad=True, kink_margin=1e-4, dtype=torch.float64):
    """-> (out, gradients of <out, G_eff>, G_eff float32); see _gat.  dtype=float32 with kink_margin=None: the reference's
    formulation differentiated op for op at the reference's own width, G taken as it is (an already-filtered G_eff)."""
    return _gat(x32, ei, wq, bq, wk, bk, wv, bias, num_heads, G, x_grad, dtype, kink_margin)


def _gat(x32, ei, wq, bq, wk, bk, wv, bias, num_heads, G, x_grad, dtype, kink_margin):
    """relu(concat_h(softmax_row(<Q, K> / sqrt(d)) V) + b) with Q = relu(x Wq + bq), K = relu(x Wk + bk), V = x W and the N
    self-loop edges appended AFTER the input edges (nn/conv/gat.py:40-122; softmax = nn/kernel/segment.py:26-33: row maximum
    under stop_gradient, 1e-8 added to the denominator).  Heads are independent given the gradient entering the concat: one
    forward pass without a graph fixes the final ReLU's mask, then each head is rebuilt with its graph, differentiated into the
    (leaf) Q / K / V and freed; the projections are differentiated last."""
    dev = x32.device
    n = int(x32.shape[0])
    g = SortedEdges(ei, n, self_loops=True)
    x = torch.as_tensor(x32).to(device=dev, dtype=dtype).clone().requires_grad_(x_grad and G is not None)
    Wq, Bq, Wk, Bk, Wv, b = (torch.as_tensor(t).to(device=dev, dtype=dtype).clone().requires_grad_(G is not None)
                             for t in (wq, bq, wk, bk, wv, bias))
    Q0, K0, V0 = torch.relu(x @ Wq + Bq), torch.relu(x @ Wk + Bk), x @ Wv
    Q, K, V = (t.detach().requires_grad_(G is not None) for t in (Q0, K0, V0))
    H = num_heads
    d, dv = int(Q.shape[1]) // H, int(V.shape[1]) // H

    def head(h):
        s = (Q[:, h * d:(h + 1) * d][g.row] * K[:, h * d:(h + 1) * d][g.col]).sum(-1) / math.sqrt(d)      # gat.py:78-79
        m = g.seg_max(s.detach(), -math.inf)
        p = torch.exp(s - m[g.row])
        a = p / (g.seg_sum(p) + 1e-8)[g.row]
        return g.seg_sum(a[:, None] * V[:, h * dv:(h + 1) * dv][g.col])

    with torch.no_grad():
        pre = torch.cat([head(h) for h in range(H)], dim=1) + b
    if G is None:
        return torch.relu(pre), None, None
    if kink_margin is None:
        out, G_eff = torch.relu(pre), G.to(dtype)
    else:
        out, G_eff = _outer_relu(pre, G, kink_margin)
    gpre = G_eff * (pre > 0)
    for h in range(H):
        o = head(h)
        o.backward(gpre[:, h * dv:(h + 1) * dv])
        del o
    torch.autograd.backward([Q0, K0, V0], [Q.grad, K.grad, V.grad])
    grads = {"x": x.grad, "query_kernel": Wq.grad, "query_bias": Bq.grad, "key_kernel": Wk.grad, "key_bias": Bk.grad,
             "kernel": Wv.grad, "bias": gpre.sum(0)}
    return out, grads, G_eff.float()
