# coding=utf-8
"""Graph-level readouts as layer objects — the working form of tf_geometric.layers.{MeanPool, SumPool, MaxPool, MinPool}
(layers/pool/common_pool.py: there the four subclasses derive from tf.keras.Model and hand the pooling function to
keras' constructor, so they cannot be instantiated; the intended behaviour is CommonPool's, :6-19).
inputs = [x, node_graph_index] or [x, node_graph_index, num_graphs]; one segment-reduce launch each."""
from ...nn.pool.common_pool import mean_pool, sum_pool, max_pool, min_pool
from .._base import Layer


class CommonPool(Layer):
    def __init__(self, pool_func, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.pool_func = pool_func

    def build(self, input_shapes):
        """Readouts own no weights."""

    def call(self, inputs, training=None, mask=None):
        x, node_graph_index = inputs[0], inputs[1]
        num_graphs = inputs[2] if len(inputs) > 2 else None
        return self.pool_func(x, node_graph_index, num_graphs)


def _readout(name, func):
    def __init__(self, *args, **kwargs):
        CommonPool.__init__(self, func, *args, **kwargs)
    return type(name, (CommonPool,), {"__init__": __init__, "__doc__": "{} over the nodes of each graph.".format(func.__name__)})


MeanPool = _readout("MeanPool", mean_pool)
SumPool = _readout("SumPool", sum_pool)
MaxPool = _readout("MaxPool", max_pool)
MinPool = _readout("MinPool", min_pool)
