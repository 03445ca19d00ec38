# coding=utf-8
"""User-defined message passing as a layer: subclass, override `map` / `reduce` / `update`, call it like any other layer.

API counterpart of tf_geometric.layers.MapReduceGNN (layers/kernel/map_reduce.py:6-40).  Arbitrary Python hooks cannot be
fused into the HIP kernels, so a subclass runs the generic route of `aggregate_neighbors` (gather on the device, the
user's callables on torch tensors); returning one of the library's own mappers / reducers / updaters from the
`get_*` methods instead selects the fused single-launch path (nn/kernel/map_reduce.py)."""
from ...nn.kernel.map_reduce import aggregate_neighbors
from .._base import Layer


class MapReduceGNN(Layer):
    """Hooks (same names and argument order as the reference):
        map(repeated_x, neighbor_x, edge_weight=None)      -> per-edge message
        reduce(neighbor_msg, node_index, num_nodes=None)   -> per-destination reduction
        update(x, reduced_neighbor_msg)                    -> new node features
    """

    def map(self, repeated_x, neighbor_x, edge_weight=None):
        raise NotImplementedError("{}.map is not defined".format(type(self).__name__))

    def reduce(self, neighbor_msg, node_index, num_nodes=None):
        raise NotImplementedError("{}.reduce is not defined".format(type(self).__name__))

    def update(self, x, reduced_neighbor_msg):
        raise NotImplementedError("{}.update is not defined".format(type(self).__name__))

    # the reference hands closures to aggregate_neighbors; bound methods carry the same signature
    def get_mapper(self):
        return self.map

    def get_reducer(self):
        return self.reduce

    def get_updater(self):
        return self.update

    def build(self, input_shapes):
        """No weights of its own; subclasses that need some create them here with add_weight."""

    def call(self, inputs, training=None, mask=None):
        if len(inputs) == 2:
            (x, edge_index), edge_weight = inputs, None
        else:
            x, edge_index, edge_weight = inputs
        return aggregate_neighbors(x, edge_index, edge_weight, mapper=self.get_mapper(), reducer=self.get_reducer(),
                                   updater=self.get_updater())
