# coding=utf-8
"""Functional API — same names as tf_geometric.nn for the message-passing hot path."""
from .kernel.map_reduce import (aggregate_neighbors, identity_mapper, neighbor_count_mapper, gcn_mapper,
                                sum_reducer, mean_reducer, max_reducer, sum_updater, identity_updater)
from .kernel.segment import segment_softmax, segment_count
from .conv.gcn import (gcn, gcn_norm_adj, gcn_norm_edge, gcn_cache_normed_edge, gcn_build_cache_by_adj, gcn_build_cache_for_graph,
                       compute_cache_key)
from .conv.gat import gat
from .conv.graph_sage import (mean_graph_sage, sum_graph_sage, gcn_graph_sage, mean_pool_graph_sage,
                              max_pool_graph_sage)
from .conv.propagation import gin, sgc, tagcn, appnp, ssgc, chebynet, le_conv, chebynet_norm_edge
from .pool import mean_pool, sum_pool, max_pool, min_pool, topk_pool
