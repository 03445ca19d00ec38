# coding=utf-8
"""Multi-head graph attention as a layer object: owns the three projection matrices and their biases, forwards to the
fused functional `gat` (tf_geometric_amd/nn/conv/gat.py).  Keeps the constructor keywords, weight names and call
signature of tf_geometric.layers.GAT (layers/conv/gat.py:9-20, 60-101) so reference checkpoints load by name."""
from ...activations import relu
from ...nn.conv.gat import gat
from .._base import Layer

# weight name -> (which width it projects to, initializer); widths are resolved in build()
_PROJECTIONS = (("query_kernel", "attention", "glorot_uniform"), ("query_bias", "attention_bias", "zeros"),
                ("key_kernel", "attention", "glorot_uniform"), ("key_bias", "attention_bias", "zeros"))


class GAT(Layer):
    """units: output width; attention_units: width of Q and K (defaults to units); num_heads heads.
    split_value_heads=True: V [F, units] is cut into heads and the heads are concatenated;
    False: V is [F, units * num_heads] and the heads are averaged.
    query_activation / key_activation default to relu as in the reference; edge_drop_rate is the dropout rate of
    the attention weights (active only under training=True).  kernel_regularizer / bias_regularizer feed `layer.losses`
    (one term per regularised weight, as keras collects them; the training loop adds them to its loss)."""

    def __init__(self, units, attention_units=None, activation=None, use_bias=True, num_heads=1,
                 split_value_heads=True, query_activation=relu, key_activation=relu, edge_drop_rate=0.0,
                 kernel_regularizer=None, bias_regularizer=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.units, self.num_heads = int(units), int(num_heads)
        self.attention_units = self.units if attention_units is None else int(attention_units)
        if self.num_heads < 1 or self.attention_units % self.num_heads:
            raise ValueError("attention_units ({}) must be a positive multiple of num_heads ({})".format(
                self.attention_units, self.num_heads))
        self.split_value_heads, self.use_bias = bool(split_value_heads), bool(use_bias)
        self.activation, self.query_activation, self.key_activation = activation, query_activation, key_activation
        self.edge_drop_rate = float(edge_drop_rate)
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        for name in ("query_kernel", "query_bias", "key_kernel", "key_bias", "kernel", "bias"):
            setattr(self, name, None)                       # created lazily from the first input's width

    def build(self, input_shapes):
        num_features = int(input_shapes[0][-1])
        shapes = {"attention": [num_features, self.attention_units], "attention_bias": [self.attention_units]}
        for name, kind, init in _PROJECTIONS:
            setattr(self, name, self.add_weight(name, shapes[kind], init))
        value_width = self.units if self.split_value_heads else self.units * self.num_heads
        self.kernel = self.add_weight("kernel", [num_features, value_width], "glorot_uniform")
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], "zeros")

    def call(self, inputs, training=None, mask=None, cache=None):
        """inputs = [x, edge_index] or [x, edge_index, edge_weight]; a third element is ignored, as in the reference
        (attention defines the weights).  cache: the graph's dict (CSR plan)."""
        x, edge_index = inputs[0], inputs[1]
        return gat(x, edge_index, self.query_kernel, self.query_bias, self.query_activation,
                   self.key_kernel, self.key_bias, self.key_activation, self.kernel, self.bias, self.activation,
                   num_heads=self.num_heads, split_value_heads=self.split_value_heads,
                   edge_drop_rate=self.edge_drop_rate, training=bool(training), cache=cache)
