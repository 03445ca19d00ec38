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
