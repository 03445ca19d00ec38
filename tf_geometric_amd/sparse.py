# coding=utf-8
"""Minimal COO SparseMatrix with the tf_sparse surface the hot path touches:
matmul / @, segment_sum, segment_softmax, add_diag, dropout, transpose — all on the HIP kernels.

tf_sparse itself is an external, un-vendored dependency of the reference (setup.py:25); the semantics here
are the ones the reference's call sites rely on (nn/conv/gcn.py:72-98,262,280; nn/conv/gat.py:83-89)."""
import torch

from . import _lib as L
from .plan import CsrPlan, segment_reduce


class SparseMatrix(object):
    def __init__(self, index, value=None, shape=None):
        self.index = L.as_i32(index)
        if self.index.numel() == 0:
            self.index = self.index.reshape(2, 0)
        E = int(self.index.shape[1])
        self._has_value = value is not None
        self.value = L.as_f32(value) if value is not None else torch.ones(E, dtype=torch.float32,
                                                                          device=self.index.device)
        if shape is None:
            n = int(self.index.max().item()) + 1 if E else 0
            shape = [n, n]
        self._shape = [int(shape[0]), int(shape[1])]
        self._plan = None
        self._value_csr = None

    @property
    def shape(self):
        return self._shape

    @property
    def plan(self):
        if self._plan is None:
            self._plan = CsrPlan.build(self.index, self._shape[0], self._shape[1])
        return self._plan

    @property
    def value_csr(self):
        if self._value_csr is None:
            self._value_csr = self.plan.edge_attr_to_csr(self.value)
        return self._value_csr

    def with_value(self, value):
        m = SparseMatrix(self.index, value, self._shape)
        m._plan = self._plan
        return m

    def matmul(self, h, num_or_size_splits=None):
        """out[r] = sum_{e: row_e = r} value_e * h[col_e]; duplicates sum. num_or_size_splits only bounds memory in
        the reference (nn/conv/gcn.py:274-280) and never changes the result: nothing [E,F]-sized exists here."""
        h = L.as_f32(h)
        squeeze = h.dim() == 1
        if squeeze:
            h = h.unsqueeze(1)
        from . import autograd as AG
        if AG.needs_grad(h, self.value):
            w = self.value[self.plan.perm.long()] if AG.needs_grad(self.value) else self.value_csr
            out = AG.aggregate(self.plan, h, L.SUM, w)
        else:
            out = segment_reduce(self.plan, h, L.SUM, w_csr=self.value_csr)
        return out[:, 0] if squeeze else out

    def __matmul__(self, h):
        return self.matmul(h)

    def segment_sum(self, axis=-1):
        """axis=-1/1: row sums; axis=0: column sums (nn/conv/gcn.py:80,88)."""
        lib = L.require_gpu()
        if axis in (-1, 1):
            plan, w = self.plan, self.value_csr
        elif axis == 0:
            plan = self.plan.transposed()
            w = plan.edge_attr_to_csr(self.value)
        else:
            raise ValueError("axis must be -1, 0 or 1")
        deg = torch.empty(plan.n_dst, dtype=torch.float32, device=self.index.device)
        L.check(lib.tfgx_segment_weight_sum_f32(L.ptr(plan.row_ptr), L.ptr(w), plan.n_dst, 0.0, L.ptr(deg),
                                                L.stream_ptr()), "tfgx_segment_weight_sum_f32")
        return deg

    def segment_softmax(self, axis=-1):
        if axis not in (-1, 1):
            raise NotImplementedError("segment_softmax is implemented for axis=-1 (rows), the only axis the "
                                      "reference uses (nn/conv/gat.py:83-84)")
        lib = L.require_gpu()
        plan = self.plan
        from . import autograd as AG
        if AG.needs_grad(self.value):
            return self.with_value(AG.segment_softmax(plan, self.index[0], self.value))
        out = torch.empty_like(self.value)
        if plan.num_edges:
            L.edge_softmax(plan, self.value.contiguous(), 1, out)
        return self.with_value(out)

    def add_diag(self, weight):
        """A + weight*I as appended (i, i, weight) entries (duplicates sum under matmul / segment_sum)."""
        n = min(self._shape)
        ar = torch.arange(n, dtype=torch.int32, device=self.index.device)
        index = torch.cat([self.index, torch.stack([ar, ar])], dim=1)
        value = torch.cat([self.value, torch.full((n,), float(weight), dtype=torch.float32, device=ar.device)])
        return SparseMatrix(index, value, self._shape)

    def dropout(self, rate, training=False):
        """Values kept with probability 1 - rate and rescaled by 1 / (1 - rate) when training (tf.nn.dropout on
        .value); the index — and therefore the cached CSR plan — is shared with self."""
        if not training or rate <= 0.0:
            return self
        if not rate < 1.0:
            raise Exception("dropout rate must be in [0, 1)")
        value = self.value * ((torch.rand_like(self.value) >= rate).to(torch.float32) * (1.0 / (1.0 - float(rate))))
        out = SparseMatrix(self.index, value, self._shape)
        out._plan = getattr(self, "_plan", None)
        return out

    def transpose(self):
        return SparseMatrix(torch.stack([self.index[1], self.index[0]]), self.value,
                            [self._shape[1], self._shape[0]])

    def to_dense(self):
        eye = torch.eye(self._shape[1], dtype=torch.float32, device=self.index.device)
        return self.matmul(eye)


def sparse_features(x):
    """x as a SparseMatrix when it is one (or a torch sparse COO tensor), else None — the reference's
    isinstance(x, tf.sparse.SparseTensor) test (nn/conv/gcn.py:269, gat.py:47, sgc.py:31, appnp.py:64 ...)."""
    if isinstance(x, SparseMatrix):
        return x
    if isinstance(x, torch.Tensor) and x.is_sparse:
        x = x.coalesce()
        return SparseMatrix(x.indices().to(torch.int32), x.values(), list(x.shape))
    return None


def sparse_dense_matmul(xs, kernel, bias=None, act=L.ACT_NONE):
    """act(xs @ kernel + bias) for sparse node features (tf.sparse.sparse_dense_matmul at the call sites above): a
    gather-scale-segment-sum over the nonzeros of xs with the KERNEL rows as the source table — the same HIP kernel as
    the neighbour aggregation.  Differentiable wrt kernel / bias when they are being tracked."""
    from . import autograd as AG
    k = L.as_f32(kernel)
    b = None if bias is None else L.as_f32(bias)
    if AG.needs_grad(k, b):
        h = AG.aggregate(xs.plan, k, L.SUM, xs.value_csr)
        if b is not None:
            h = h + b
        return torch.relu(h) if act == L.ACT_RELU else h
    return segment_reduce(xs.plan, k, L.SUM, w_csr=xs.value_csr, bias=None if b is None else b.contiguous(), act=act)
