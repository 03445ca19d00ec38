# coding=utf-8
"""aggregate_neighbors and its mapper / reducer / updater vocabulary on the HIP backend.

Mirror of tf_geometric/nn/kernel/map_reduce.py (same names, same argument meaning).  The built-in
(mapper, reducer, updater) triples run as ONE fused kernel launch (gather + scale + segment reduce +
update); an arbitrary Python mapper takes the generic route: gather kernels -> user mapper on device
tensors -> segment-reduce kernel over the explicit messages.
"""
import torch

from ... import _lib as L
from ...plan import CsrPlan, segment_reduce
from ... import autograd as AG


def identity_mapper(repeated_x, neighbor_x, edge_weight=None):
    return neighbor_x                                       # map_reduce.py:7-8


def neighbor_count_mapper(repeated_x, neighbor_x, edge_weight=None):
    return torch.ones([neighbor_x.shape[0], 1], dtype=torch.float32, device=neighbor_x.device)   # :11-12


def gcn_mapper(repeated_x, neighbor_x, edge_weight=None):
    return neighbor_x * L.as_f32(edge_weight).unsqueeze(1)  # nn/conv/gcn.py:221-222 (None raises, as there)


def _reduce_messages(neighbor_msg, node_index, num_nodes, op):
    """unsorted_segment_{sum,mean,max}(msg, node_index, num_nodes): message i goes to row node_index[i]."""
    msg = L.as_f32(neighbor_msg)
    squeeze = msg.dim() == 1
    if squeeze:
        msg = msg.unsqueeze(1)
    ids = L.as_i32(node_index)
    if num_nodes is None:
        num_nodes = int(msg.shape[0]) if op != L.MAX else int(ids.max().item()) + 1
    # plan with col = edge id: the "gather" reads message perm[i] for CSR position i
    plan = CsrPlan.build(torch.stack([ids, torch.arange(ids.shape[0], dtype=torch.int32, device=ids.device)]),
                         int(num_nodes), max(int(msg.shape[0]), 1))
    out = AG.aggregate(plan, msg, op) if AG.needs_grad(msg) else segment_reduce(plan, msg, op)
    return out[:, 0] if squeeze else out


def sum_reducer(neighbor_msg, node_index, num_nodes=None):
    return _reduce_messages(neighbor_msg, node_index, num_nodes, L.SUM)    # :15-16


def mean_reducer(neighbor_msg, node_index, num_nodes=None):
    return _reduce_messages(neighbor_msg, node_index, num_nodes, L.MEAN)   # :27-28


def max_reducer(neighbor_msg, node_index, num_nodes=None):
    return _reduce_messages(neighbor_msg, node_index, num_nodes, L.MAX)    # :38-42 (num_nodes None -> max+1)


def sum_updater(x, reduced_neighbor_msg):
    return x + reduced_neighbor_msg                         # :19-20


def identity_updater(x, reduced_neighbor_msg):
    return reduced_neighbor_msg                             # :23-24


_REDUCER_OPS = {sum_reducer: L.SUM, mean_reducer: L.MEAN, max_reducer: L.MAX}


def aggregate_neighbors(x, edge_index, edge_weight=None, mapper=identity_mapper,
                        reducer=sum_reducer, updater=sum_updater, num_nodes=None, cache=None):
    """
    updater(x, reducer(mapper(x[row], x[col], edge_weight), row, num_nodes))   — reference :45-73.

    :param x: [num_nodes, num_features] node features
    :param edge_index: [2, num_edges]; edge_index[0] aggregates, edge_index[1] is the neighbour
    :param mapper: (features_of_node, features_of_neighbor_node, edge_weight) => neighbor_msg
    :param reducer: (neighbor_msg, node_index, num_nodes) => reduced_neighbor_msg
    :param updater: (features_of_node, reduced_neighbor_msg) => aggregated_node_features
    :param num_nodes: number of nodes (defaults to x.shape[0])
    :param cache: optional dict holding the per-graph CSR plan (extension; the reference has no plan to cache)
    """
    L.require_gpu()
    x = L.as_f32(x)
    ei = L.as_i32(edge_index)
    if ei.shape[0] == 0:                                    # :57 tests dimension 0 only: "no edges" given as []
        return x
    # a [2, 0] edge_index goes on, as in the reference: every segment is empty (0 for sum/mean, float32 lowest for max)
    n = int(x.shape[0]) if num_nodes is None else int(num_nodes)
    fused = (mapper in (identity_mapper, gcn_mapper)) and (reducer in _REDUCER_OPS) and \
            (updater in (sum_updater, identity_updater)) and (updater is identity_updater or n == int(x.shape[0]))
    if fused:
        if mapper is gcn_mapper and edge_weight is None:
            raise TypeError("gcn_mapper needs edge_weight (tf.expand_dims(None) in the reference, gcn.py:222)")
        plan = CsrPlan.from_cache(ei, n, int(x.shape[0]), cache)
        w_csr = AG.edge_attr_csr(plan, edge_weight, cache) if mapper is gcn_mapper else None
        if AG.needs_grad(x, edge_weight):        # training route: kernels with a backward (autograd.py)
            red = AG.aggregate(plan, x, _REDUCER_OPS[reducer], w_csr)
            return x + red if updater is sum_updater else red
        return segment_reduce(plan, x, _REDUCER_OPS[reducer], w_csr=w_csr,
                              add_x=x if updater is sum_updater else None)
    # generic route: explicit gathers, user mapper, HIP reducer
    repeated_x = AG.gather(x, ei[0])                        # :62
    neighbor_x = AG.gather(x, ei[1])                        # :63
    neighbor_msg = mapper(repeated_x, neighbor_x, edge_weight=edge_weight)     # :65
    reduced_msg = reducer(neighbor_msg, ei[0], num_nodes=n)                    # :70
    return updater(x, reduced_msg)                          # :71
