# coding=utf-8
"""GAT layer — drop-in for tf_geometric.layers.GAT (reference: layers/conv/gat.py)."""
from ...activations import relu
from ...nn.conv.gat import gat
from .._base import Layer


class GAT(Layer):
    """Constructor arguments as layers/conv/gat.py:9-20 (query/key activations default to relu)."""

    def __init__(self, units, attention_units=None, activation=None, use_bias=True, num_heads=1,
                 split_value_heads=True, query_activation=relu, key_activation=relu, edge_drop_rate=0.0,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units = units
        self.attention_units = units if attention_units is None else attention_units
        self.edge_drop_rate = edge_drop_rate
        self.query_kernel = None
        self.query_bias = None
        self.query_activation = query_activation
        self.key_kernel = None
        self.key_bias = None
        self.key_activation = key_activation
        self.kernel = None
        self.bias = None
        self.activation = activation
        self.use_bias = use_bias
        self.num_heads = num_heads
        self.split_value_heads = split_value_heads
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer

    def build(self, input_shapes):
        f = input_shapes[0][-1]
        self.query_kernel = self.add_weight("query_kernel", [f, self.attention_units], "glorot_uniform")   # :64-65
        self.query_bias = self.add_weight("query_bias", [self.attention_units], "zeros")
        self.key_kernel = self.add_weight("key_kernel", [f, self.attention_units], "glorot_uniform")       # :69-70
        self.key_bias = self.add_weight("key_bias", [self.attention_units], "zeros")
        width = self.units if self.split_value_heads else self.units * self.num_heads                       # :74-79
        self.kernel = self.add_weight("kernel", [f, width], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")                                      # :81-83

    def call(self, inputs, training=None, mask=None, cache=None):
        """:param inputs: [x, edge_index] or [x, edge_index, edge_weight] — edge_weight is ignored (:88-92)."""
        x, edge_index = inputs[0], inputs[1]
        return gat(x, edge_index, self.query_kernel, self.query_bias, self.query_activation,
                   self.key_kernel, self.key_bias, self.key_activation, self.kernel, self.bias, self.activation,
                   num_heads=self.num_heads, split_value_heads=self.split_value_heads,
                   edge_drop_rate=self.edge_drop_rate, training=bool(training), cache=cache)
