# coding=utf-8
"""Segment ops by destination — HIP-backed counterparts of tf_geometric/nn/kernel/segment.py."""
import torch

from ... import _lib as L
from ...plan import CsrPlan
from ... import autograd as AG


def segment_softmax(data, segment_ids, num_segments):
    """exp(d - segmax) / (segsum + 1e-8) grouped by segment_ids (reference: nn/kernel/segment.py:26-33).
    data: [E] or [E, k]; returned in the caller's order."""
    lib = L.require_gpu()
    d = L.as_f32(data).contiguous()
    ids = L.as_i32(segment_ids)
    E = int(ids.shape[0])
    squeeze = d.dim() == 1
    H = 1 if squeeze else int(d.shape[1])
    plan = CsrPlan.build(torch.stack([ids, torch.zeros_like(ids)]), int(num_segments), 1)
    if AG.needs_grad(d):
        return AG.segment_softmax(plan, ids, d)
    out = torch.empty_like(d)
    if E:
        L.edge_softmax(plan, d, H, out)
    return out


def segment_count(index, num_segments=None):
    """Number of entries per segment (reference: nn/kernel/segment.py:36-40)."""
    ids = L.as_i32(index)
    if num_segments is None:
        num_segments = int(ids.max().item()) + 1
    plan = CsrPlan.build(torch.stack([ids, torch.zeros_like(ids)]), int(num_segments), 1)
    return plan.in_degree().to(ids.dtype)


# ---------------------------------------------------------------------------------------------------------------------
# The TF1 route of the reference's max reducer / max-min readouts (nn/kernel/map_reduce.py:31-36, nn/pool/common_pool.py:
# 24-37): `segment_op_with_pad(tf.math.segment_max, data, ids, n)`.  The callables it is handed are TensorFlow's SORTED
# segment ops; their counterparts here are segment_sum / segment_mean / segment_max / segment_min below.
# ---------------------------------------------------------------------------------------------------------------------
def _rows(data):
    d = L.as_f32(data)
    return (d.reshape(-1, 1), True) if d.dim() == 1 else (d.reshape(int(d.shape[0]), -1), False)


def _integer_dtype(data):
    """torch integer dtype of `data` (a torch tensor or anything numpy understands), None for floating point."""
    if isinstance(data, torch.Tensor):
        return None if data.dtype.is_floating_point else (data.dtype if data.dtype != torch.bool else torch.int32)
    import numpy as np
    kind = np.asarray(data).dtype.kind
    if kind in "iub":
        return {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}.get(np.asarray(data).dtype.itemsize, torch.int64) \
            if kind == "i" else (torch.int64 if kind == "u" else torch.int32)
    return None


def _segment(data, segment_ids, num_segments, op):
    """One launch of the segment-reduce kernel over rows grouped by id; a segment without rows holds 0 (the sorted TF ops'
    documented value for an empty segment — NOT float32 lowest, which is what the unsorted max holds)."""
    int_dtype = _integer_dtype(data)              # the sorted TF ops (and segment_op_with_pad's pads) keep the data dtype
    d = L.as_f32(data)
    d2, squeeze = _rows(d)
    ids = L.as_i32(segment_ids)
    n_rows = int(ids.shape[0])
    if int(d2.shape[0]) != n_rows:
        raise ValueError("segment op: data has {} rows, segment_ids {}".format(int(d2.shape[0]), n_rows))
    n = int(num_segments)
    if n_rows and int(ids.min().item()) < 0:
        raise ValueError("segment op: negative segment id {}".format(int(ids.min().item())))      # TF: InvalidArgumentError
    from ...plan import segment_reduce
    plan = CsrPlan.build(torch.stack([ids, torch.arange(n_rows, dtype=torch.int32, device=ids.device)]), n, max(n_rows, 1))
    kind, flip = {"sum": (L.SUM, False), "mean": (L.MEAN, False), "max": (L.MAX, False), "min": (L.MAX, True)}[op]
    src = -d2 if flip else d2
    out = AG.aggregate(plan, src, kind) if AG.needs_grad(src) else segment_reduce(plan, src, kind)
    if flip:
        out = -out
    if kind == L.MAX:
        out = torch.where((plan.in_degree() == 0).unsqueeze(-1), torch.zeros((), dtype=out.dtype, device=out.device), out)
    if int_dtype is not None:
        # integer data (counts, ids): the kernel computes in float32, which is exact only below 2^24 — checked, not assumed
        if op == "mean":
            raise TypeError("segment_mean of integer data is not supported (float32 kernel): cast to float32 first")
        terms = int(plan.in_degree().max().item()) if (n_rows and op == "sum") else 1
        bound = (float(d2.abs().max().item()) if n_rows else 0.0) * max(terms, 1)
        if bound >= 2.0 ** 24:
            raise TypeError("segment_{} of {} data: |value| x segment length reaches {:.3g} >= 2^24, beyond what the float32 "
                            "kernel represents exactly".format(op, int_dtype, bound))
        out = out.to(int_dtype)
    return out.reshape(n) if squeeze else out.reshape((n,) + tuple(d.shape[1:]))


def _sorted_segment_op(op):
    def segment_op(data, segment_ids, name=None):
        """tf.math.segment_{} : ids ascending, output rows = last id + 1, an id that does not occur yields 0."""
        ids = L.as_i32(segment_ids)
        if int(ids.shape[0]) == 0:
            d = L.as_f32(data)
            return torch.zeros((0,) + tuple(d.shape[1:]), dtype=_integer_dtype(data) or torch.float32, device=d.device)
        if bool((ids[1:] < ids[:-1]).any().item()):
            raise ValueError("segment ids are not increasing")            # TF: InvalidArgumentError
        return _segment(data, ids, int(ids[-1].item()) + 1, op)
    segment_op.__doc__ = segment_op.__doc__.format(op)
    segment_op.__name__ = "segment_" + op
    segment_op._tfgx_segment_kind = op
    return segment_op


segment_sum, segment_mean = _sorted_segment_op("sum"), _sorted_segment_op("mean")
segment_max, segment_min = _sorted_segment_op("max"), _sorted_segment_op("min")


def segment_op_with_pad(segment_op, data, segment_ids, num_segments):
    """Reference: nn/kernel/segment.py:5-23 — sort the rows by segment id, apply a SORTED segment op (it returns last id + 1
    rows), pad with zero rows up to num_segments.  For this module's own segment_sum / mean / max / min that is ONE launch of
    the segment-reduce kernel on the unsorted rows (no sort, no gather, no concat); any other callable
    (sorted_data, sorted_ids) -> [<= num_segments, ...] is run as the reference runs it."""
    n = int(num_segments)
    kind = getattr(segment_op, "_tfgx_segment_kind", None)
    ids = L.as_i32(segment_ids)
    if kind is not None:
        if int(ids.shape[0]) and int(ids.max().item()) >= n:
            raise ValueError("segment_op_with_pad: segment id {} >= num_segments {}".format(int(ids.max().item()), n))
        return _segment(data, ids, n, kind)
    d = L.as_f32(data)
    order = torch.argsort(ids.long(), stable=True)                                       # :7
    reduced = segment_op(d[order], ids[order])                                           # :8-11
    pads = torch.zeros((n - int(reduced.shape[0]),) + tuple(d.shape[1:]), dtype=reduced.dtype, device=reduced.device)   # :12-18
    return torch.cat([reduced, pads], dim=0)                                             # :19-23
