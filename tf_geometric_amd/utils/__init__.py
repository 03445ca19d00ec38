# coding=utf-8
"""The one piece of tf_geometric.utils the hot path touches: add_self_loop_edge (GAT prologue)."""
import numpy as np
import torch

from .. import _lib as L


def add_self_loop_edge(edge_index, num_nodes, edge_weight=None, fill_weight=1.0):
    """APPEND the N diagonal edges after the input edges; existing self-loops are kept (duplicates).
    Reference: tf_geometric/utils/graph_utils.py:350-366.  (The fused GAT / GCN kernels keep this edge implicit;
    this explicit form exists for user code that calls it directly.)  numpy in -> numpy out, tensor in -> tensor out."""
    is_np = not isinstance(edge_index, torch.Tensor)
    if is_np:
        diag = np.stack([np.arange(num_nodes, dtype=np.int32)] * 2, axis=0)
        ei = np.concatenate([np.asarray(edge_index, dtype=np.int32).reshape(2, -1), diag], axis=1)
        ew = None
        if edge_weight is not None:
            ew = np.concatenate([np.asarray(edge_weight, dtype=np.float32),
                                 np.full([num_nodes], fill_weight, dtype=np.float32)])
        return ei, ew
    ar = torch.arange(num_nodes, dtype=torch.int32, device=edge_index.device)
    ei = torch.cat([edge_index.to(torch.int32), torch.stack([ar, ar])], dim=1)
    ew = None
    if edge_weight is not None:
        ew = torch.cat([L.as_f32(edge_weight, edge_index.device),
                        torch.full((num_nodes,), float(fill_weight), dtype=torch.float32, device=edge_index.device)])
    return ei, ew
