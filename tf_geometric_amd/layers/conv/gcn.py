# coding=utf-8
"""GCN layer — drop-in for tf_geometric.layers.GCN (reference: layers/conv/gcn.py)."""
import warnings

from ..._lib import as_f32
from ...nn.conv.gcn import gcn, gcn_build_cache_for_graph, gcn_build_cache_by_adj
from ...sparse import SparseMatrix
from ...plan import CACHE_KEY_PLAN
from .._base import Layer


class GCN(Layer):
    """Graph Convolutional Layer; constructor arguments as layers/conv/gcn.py:32-40."""

    def __init__(self, units, activation=None, use_kernel=True, use_bias=True,
                 norm="both", add_self_loop=True, sym=True, renorm=True, improved=False,
                 edge_drop_rate=0.0, num_splits=None, num_or_size_splits=None,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units = units
        self.activation = activation
        self.use_kernel = use_kernel
        self.use_bias = use_bias
        self.edge_drop_rate = edge_drop_rate
        self.kernel = None
        self.bias = None
        self.norm = norm
        self.add_self_loop = add_self_loop
        self.sym = sym
        self.renorm = renorm
        self.improved = improved
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        if num_splits is not None and num_or_size_splits is not None:
            raise Exception("cannot provide both num_splits and num_or_size_splits for GCN")   # :82-83
        self.num_splits = num_splits
        self.num_or_size_splits = num_or_size_splits

    def build(self, input_shapes):
        num_features = input_shapes[0][-1]
        if self.num_splits is not None:
            # as the reference's build (layers/conv/gcn.py:21-23 -> utils/tf_sparse_utils.py:71-90): the split is computed
            # here and an impossible num_splits raises at build time, not at the first call
            from ...dist.sharded import compute_num_or_size_splits
            num_h_features = self.units if self.use_kernel else num_features
            self.num_or_size_splits = compute_num_or_size_splits(num_h_features, self.num_splits)
        if self.use_kernel:
            self.kernel = self.add_weight("kernel", [num_features, self.units], "glorot_uniform")     # :26-27
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units if self.use_kernel else num_features], "zeros")  # :29-30

    def build_cache_by_adj(self, sparse_adj, override=False, cache=None):
        return gcn_build_cache_by_adj(sparse_adj, self.norm, self.add_self_loop, self.sym, self.renorm,
                                      self.improved, override=override, cache=cache)

    def build_cache_for_graph(self, graph, override=False):
        gcn_build_cache_for_graph(graph, self.norm, self.add_self_loop, self.sym, self.renorm, self.improved,
                                  override=override)

    def cache_normed_edge(self, graph, override=False):
        warnings.warn("'GCN.cache_normed_edge(graph, override)' is deprecated, use "
                      "'GCN.build_cache_for_graph(graph, override)' instead", DeprecationWarning)
        return self.build_cache_for_graph(graph, override=override)

    def call(self, inputs, cache=None, split=True, training=None, mask=None):
        """
        :param inputs: [x, sparse_adj], [x, edge_index] or [x, edge_index, edge_weight]   (:129-148)
        :param cache: dict caching the normalised adjacency (and the CSR plan) of ONE graph
        :return: [num_nodes, units]
        """
        if isinstance(inputs[1], SparseMatrix):
            x, sparse_adj = inputs
        else:
            x = inputs[0]
            edge_index = inputs[1]
            edge_weight = inputs[2] if len(inputs) == 3 else None
            num_nodes = int(x.shape[0])
            key = "tfgx_gcn_adj"
            sparse_adj = cache.get(key) if cache is not None else None
            if sparse_adj is None:
                sparse_adj = SparseMatrix(edge_index, value=edge_weight, shape=[num_nodes, num_nodes])
                if cache is not None:
                    if cache.get(CACHE_KEY_PLAN) is not None:      # one CSR plan per graph, shared by all layers
                        sparse_adj._plan = cache[CACHE_KEY_PLAN]
                    else:
                        cache[CACHE_KEY_PLAN] = sparse_adj.plan
                    cache[key] = sparse_adj
        x_in = x if (isinstance(x, SparseMatrix) or getattr(x, "is_sparse", False)) else as_f32(x)   # sparse x: :269-270
        return gcn(x_in, sparse_adj, self.kernel, self.bias, activation=self.activation,
                   norm=self.norm, add_self_loop=self.add_self_loop, sym=self.sym, renorm=self.renorm,
                   improved=self.improved, edge_drop_rate=self.edge_drop_rate,
                   num_or_size_splits=self.num_or_size_splits if split else None,
                   training=bool(training), cache=cache)
