# coding=utf-8
"""Graph-level readouts as segment reduces over node_graph_index — the same kernel as the
neighbour aggregation, with "destination" = graph id.  Reference: tf_geometric/nn/pool/common_pool.py:7-52."""
import torch

from ... import _lib as L
from ...plan import CsrPlan, segment_reduce
from ... import autograd as AG


def _pool(x, node_graph_index, num_graphs, op):
    x = L.as_f32(x)
    ids = L.as_i32(node_graph_index)
    if num_graphs is None:
        num_graphs = int(ids.max().item()) + 1                      # :8-9
    n = int(ids.shape[0])
    plan = CsrPlan.build(torch.stack([ids, torch.arange(n, dtype=torch.int32, device=ids.device)]), int(num_graphs),
                         max(n, 1))
    if AG.needs_grad(x):         # graph classification trains THROUGH the readout (the reference's TF ops all have gradients)
        return plan, x, AG.aggregate(plan, x, op)
    return plan, x, segment_reduce(plan, x, op)


def sum_pool(x, node_graph_index, num_graphs=None):
    """:17-21."""
    return _pool(x, node_graph_index, num_graphs, L.SUM)[2]


def mean_pool(x, node_graph_index, num_graphs=None):
    """sum / (count + 1e-8) — note the epsilon form, not max(count, 1) (:7-12)."""
    plan, _, s = _pool(x, node_graph_index, num_graphs, L.SUM)
    return s / (plan.in_degree().to(torch.float32).unsqueeze(-1) + 1e-8)


def max_pool(x, node_graph_index, num_graphs=None):
    """unsorted_segment_max; an empty graph holds float32 lowest (:41-45)."""
    return _pool(x, node_graph_index, num_graphs, L.MAX)[2]


def min_pool(x, node_graph_index, num_graphs=None):
    """unsorted_segment_min = -max(-x); an empty graph holds float32 max (:48-52)."""
    x = L.as_f32(x)
    return -_pool(-x, node_graph_index, num_graphs, L.MAX)[2]
