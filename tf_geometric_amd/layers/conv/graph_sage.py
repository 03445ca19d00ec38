# coding=utf-8
"""GraphSAGE layers — drop-ins for tf_geometric.layers.{Mean,Sum,GCN,MeanPool,MaxPool}GraphSage
(reference: layers/conv/graph_sage.py; weight names kept for checkpoint compatibility)."""
from ...activations import relu
from ...nn.conv.graph_sage import (mean_graph_sage, sum_graph_sage, gcn_graph_sage, mean_pool_graph_sage,
                                   max_pool_graph_sage)
from .._base import Layer


def _unpack(inputs):
    if len(inputs) == 3:
        return inputs
    x, edge_index = inputs
    return x, edge_index, None


class _SelfNeighborSage(Layer):
    _fn = None

    def __init__(self, units, activation=relu, use_bias=True, concat=True, normalize=False,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units = units
        self.activation = activation
        self.use_bias = use_bias
        self.concat = concat
        self.normalize = normalize
        if concat and (units % 2 != 0):
            raise Exception("units must be a event number if concat is True")      # :36-37
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        self.self_kernel = None
        self.neighbor_kernel = None
        self.bias = None

    def build(self, input_shape):
        f = input_shape[0][-1]
        ku = self.units // 2 if self.concat else self.units                       # :50-53
        self.self_kernel = self.add_weight("self_kernel", [f, ku], "glorot_uniform")
        self.neighbor_kernel = self.add_weight("neighbor_kernel", [f, ku], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        """:param inputs: [x, edge_index] or [x, edge_index, edge_weight]   (:63-74)"""
        x, edge_index, edge_weight = _unpack(inputs)
        return type(self)._fn(x, edge_index, edge_weight, self.self_kernel, self.neighbor_kernel, bias=self.bias,
                              activation=self.activation, concat=self.concat, normalize=self.normalize, cache=cache)


class MeanGraphSage(_SelfNeighborSage):
    """layers/conv/graph_sage.py:8-81."""
    _fn = staticmethod(mean_graph_sage)


class SumGraphSage(_SelfNeighborSage):
    """layers/conv/graph_sage.py:83-156."""
    _fn = staticmethod(sum_graph_sage)


class GCNGraphSage(Layer):
    """layers/conv/graph_sage.py:159-203."""

    def __init__(self, units, activation=relu, use_bias=True, normalize=False, kernel_regularizer=None,
                 bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units = units
        self.activation = activation
        self.use_bias = use_bias
        self.normalize = normalize
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        f = input_shape[0][-1]
        self.kernel = self.add_weight("kernel", [f, self.units], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        return gcn_graph_sage(x, edge_index, edge_weight, self.kernel, self.bias, self.activation, self.normalize,
                              cache=cache)


class _PoolSage(Layer):
    _fn = None
    _mlp_names = ("neighbor_mlp_kernel", "neighbor_mlp_bias", "neighbor_kernel")

    def __init__(self, units, activation=relu, use_bias=True, concat=True, normalize=False,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units = units
        self.activation = activation
        self.use_bias = use_bias
        self.concat = concat
        if concat and (units % 2 != 0):
            raise Exception("units must be a event number if concat is True")
        self.normalize = normalize
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        self.self_kernel = None
        self.neighbor_mlp_kernel = None
        self.neighbor_mlp_bias = None
        self.neighbor_kernel = None
        self.bias = None

    def build(self, input_shape):
        f = input_shape[0][-1]
        ku = self.units // 2 if self.concat else self.units
        n_mlp_k, n_mlp_b, n_neigh = self._mlp_names
        self.self_kernel = self.add_weight("self_kernel", [f, ku], "glorot_uniform")
        self.neighbor_mlp_kernel = self.add_weight(n_mlp_k, [f, ku * 4], "glorot_uniform")      # :248-249 / :322-323
        if self.use_bias:
            self.neighbor_mlp_bias = self.add_weight(n_mlp_b, [ku * 4], "zeros")
        self.neighbor_kernel = self.add_weight(n_neigh, [ku * 4, ku], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        return type(self)._fn(x, edge_index, edge_weight, self.self_kernel, self.neighbor_mlp_kernel,
                              self.neighbor_kernel, neighbor_mlp_bias=self.neighbor_mlp_bias, bias=self.bias,
                              activation=self.activation, concat=self.concat, normalize=self.normalize, cache=cache)


class MeanPoolGraphSage(_PoolSage):
    """layers/conv/graph_sage.py:206-281 (variables neighbor_mlp_kernel / neighbor_mlp_bias / neighbor_kernel)."""
    _fn = staticmethod(mean_pool_graph_sage)


class MaxPoolGraphSage(_PoolSage):
    """layers/conv/graph_sage.py:284-354 (variables mlp_kernel / mlp_bias / neighs_kernel, :322-328)."""
    _fn = staticmethod(max_pool_graph_sage)
    _mlp_names = ("mlp_kernel", "mlp_bias", "neighs_kernel")
