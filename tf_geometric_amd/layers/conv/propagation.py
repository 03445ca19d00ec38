# coding=utf-8
"""Layer wrappers of nn/conv/propagation.py — drop-ins for tf_geometric.layers.{GIN, SGC, TAGCN, APPNP, SSGC,
ChebyNet, LEConv} (reference: layers/conv/{gin,sgc,tagcn,appnp,ssgc,chebynet,le_conv}.py; same constructor
arguments and weight names)."""

from ...activations import relu
from ...nn.conv.propagation import gin, sgc, tagcn, appnp, ssgc, chebynet, le_conv, chebynet_norm_edge
from .._base import Layer


def _unpack(inputs):
    if len(inputs) == 3:
        return inputs
    return inputs[0], inputs[1], None


class GIN(Layer):
    """layers/conv/gin.py:11-23."""

    def __init__(self, mlp_model, eps=0, train_eps=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mlp_model = mlp_model
        self.eps = eps
        self.train_eps = train_eps

    def build(self, input_shapes):
        if self.train_eps:
            self.eps = self.add_weight("eps", [], "zeros")

    def call(self, inputs, training=None, mask=None, cache=None):
        x, edge_index, _ = _unpack(inputs)
        return gin(x, edge_index, self.mlp_model, self.eps, training=training, cache=cache)


class SGC(Layer):
    """layers/conv/sgc.py:14-49."""

    def __init__(self, units, k=1, activation=None, use_bias=True, renorm=True, improved=False,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.units, self.k, self.activation, self.use_bias = units, k, activation, use_bias
        self.renorm, self.improved = renorm, improved
        self.kernel = self.bias = None

    def build(self, input_shapes):
        self.kernel = self.add_weight("kernel", [input_shapes[0][-1], self.units], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        return sgc(x, edge_index, edge_weight, self.k, self.kernel, self.bias, self.activation, self.renorm,
                   self.improved, cache)


class TAGCN(Layer):
    """layers/conv/tagcn.py:17-57 (kernel is [F*(k+1), units])."""

    def __init__(self, units, k=3, activation=None, use_bias=True, renorm=False, improved=False,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        assert k > 0                                                     # layers/conv/tagcn.py:34
        self.units, self.k, self.activation, self.use_bias = units, k, activation, use_bias
        self.renorm, self.improved = renorm, improved
        self.kernel = self.bias = None

    def build(self, input_shapes):
        self.kernel = self.add_weight("kernel", [input_shapes[0][-1] * (self.k + 1), self.units], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        return tagcn(x, edge_index, edge_weight, self.k, self.kernel, self.bias, self.activation, self.renorm,
                     self.improved, cache)


class _MlpPropagation(Layer):
    def __init__(self, units_list=None, dense_activation=relu, activation=None, k=10, alpha=0.1, dense_drop_rate=0.0,
                 last_dense_drop_rate=0.0, edge_drop_rate=0.0, kernel_regularizer=None, bias_regularizer=None,
                 *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.units_list = units_list
        self.dense_activation, self.activation = dense_activation, activation
        self.k, self.alpha = k, alpha
        self.dense_drop_rate, self.last_dense_drop_rate, self.edge_drop_rate = dense_drop_rate, last_dense_drop_rate, edge_drop_rate
        self.kernels, self.biases = [], []

    def build(self, input_shapes):
        last = input_shapes[0][-1]
        for i, units in enumerate(self.units_list or []):
            k_name, b_name = "kernel_{}".format(i), "bias_{}".format(i)        # layers/conv/appnp.py:14-22
            setattr(self, k_name, None)
            setattr(self, b_name, None)
            setattr(self, k_name, self.add_weight(k_name, [last, units], "glorot_uniform"))
            setattr(self, b_name, self.add_weight(b_name, [units], "zeros"))
            last = units

    def _weights_lists(self):
        n = len(self.units_list or [])
        if n == 0:
            return None, None
        return ([getattr(self, "kernel_{}".format(i)) for i in range(n)],
                [getattr(self, "bias_{}".format(i)) for i in range(n)])


class APPNP(_MlpPropagation):
    """layers/conv/appnp.py:31-35."""

    def __init__(self, units_list, dense_activation=relu, activation=None, k=10, alpha=0.1, dense_drop_rate=0.0,
                 last_dense_drop_rate=0.0, edge_drop_rate=0.0, kernel_regularizer=None, bias_regularizer=None,
                 *args, **kwargs):
        super().__init__(units_list, dense_activation, activation, k, alpha, dense_drop_rate, last_dense_drop_rate,
                         edge_drop_rate, kernel_regularizer, bias_regularizer, *args, **kwargs)

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        kernels, biases = self._weights_lists()
        return appnp(x, edge_index, edge_weight, kernels, biases, self.dense_activation, self.activation, self.k,
                     self.alpha, self.dense_drop_rate, self.last_dense_drop_rate, self.edge_drop_rate, cache,
                     bool(training))


class SSGC(_MlpPropagation):
    """layers/conv/ssgc.py:35-41."""

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        kernels, biases = self._weights_lists()
        return ssgc(x, edge_index, edge_weight, kernels, biases, self.k, self.alpha, self.dense_activation,
                    self.activation, self.dense_drop_rate, self.last_dense_drop_rate, self.edge_drop_rate, cache,
                    bool(training))


class ChebyNet(Layer):
    """layers/conv/chebynet.py:17-67 (weights kernel0..kernel{k-1}, bias)."""

    def __init__(self, units, k, activation=None, use_bias=True, normalization_type="sym",
                 use_dynamic_lambda_max=False, kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        assert k >= 1                                                     # layers/conv/chebynet.py:38-39
        assert normalization_type in [None, "sym", "rw"]
        self.units, self.k, self.activation, self.use_bias = units, k, activation, use_bias
        self.normalization_type, self.use_dynamic_lambda_max = normalization_type, use_dynamic_lambda_max
        self.bias = None

    def build(self, input_shapes):
        f = input_shapes[0][-1]
        for i in range(self.k):
            name = "kernel{}".format(i)
            setattr(self, name, None)
            setattr(self, name, self.add_weight(name, [f, self.units], "glorot_uniform"))
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def build_cache_for_graph(self, graph, override=False):
        if override:
            graph.cache["chebynet_normed_edge_{}".format(self.normalization_type)] = None
        chebynet_norm_edge(graph.edge_index, int(graph.x.shape[0]), getattr(graph, "edge_weight", None),
                           self.normalization_type, self.use_dynamic_lambda_max, graph.cache)

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        kernels = [getattr(self, "kernel{}".format(i)) for i in range(self.k)]
        return chebynet(x, edge_index, edge_weight, self.k, kernels, self.bias, self.activation,
                        self.normalization_type, self.use_dynamic_lambda_max, cache)


class LEConv(Layer):
    """layers/conv/le_conv.py:34-38."""

    def __init__(self, units, activation=None, self_use_bias=True, aggr_self_use_bias=True,
                 aggr_neighbor_use_bias=False, kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.units, self.activation = units, activation
        self.self_use_bias, self.aggr_self_use_bias, self.aggr_neighbor_use_bias = \
            self_use_bias, aggr_self_use_bias, aggr_neighbor_use_bias
        self.self_kernel = self.self_bias = self.aggr_self_kernel = self.aggr_self_bias = None
        self.aggr_neighbor_kernel = self.aggr_neighbor_bias = None

    def build(self, input_shapes):
        f = input_shapes[0][-1]
        self.self_kernel = self.add_weight("self_kernel", [f, self.units], "glorot_uniform")
        if self.self_use_bias:
            self.self_bias = self.add_weight("self_bias", [self.units], "zeros")
        self.aggr_self_kernel = self.add_weight("aggr_self_kernel", [f, self.units], "glorot_uniform")
        if self.aggr_self_use_bias:
            self.aggr_self_bias = self.add_weight("aggr_self_bias", [self.units], "zeros")
        self.aggr_neighbor_kernel = self.add_weight("aggr_neighbor_kernel", [f, self.units], "glorot_uniform")
        if self.aggr_neighbor_use_bias:
            self.aggr_neighbor_bias = self.add_weight("aggr_neighbor_bias", [self.units], "zeros")

    def call(self, inputs, cache=None, training=None, mask=None):
        x, edge_index, edge_weight = _unpack(inputs)
        return le_conv(x, edge_index, edge_weight, self.self_kernel, self.self_bias, self.aggr_self_kernel,
                       self.aggr_self_bias, self.aggr_neighbor_kernel, self.aggr_neighbor_bias, self.activation, cache)
