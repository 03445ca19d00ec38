# coding=utf-8
"""MapReduceGNN — drop-in for tf_geometric.layers.MapReduceGNN (reference: layers/kernel/map_reduce.py:6-40)."""
from ...nn.kernel.map_reduce import aggregate_neighbors
from .._base import Layer


class MapReduceGNN(Layer):
    def map(self, repeated_x, neighbor_x, edge_weight=None):
        pass

    def reduce(self, neighbor_msg, node_index, num_nodes=None):
        pass

    def update(self, x, reduced_neighbor_msg):
        pass

    def get_mapper(self):
        return lambda repeated_x, neighbor_x, edge_weight=None: self.map(repeated_x, neighbor_x, edge_weight)

    def get_reducer(self):
        return lambda neighbor_msg, node_index, num_nodes=None: self.reduce(neighbor_msg, node_index, num_nodes)

    def get_updater(self):
        return lambda x, reduced_neighbor_msg: self.update(x, reduced_neighbor_msg)

    def build(self, input_shapes):
        pass

    def call(self, inputs, training=None, mask=None):
        x, edge_index, edge_weight = inputs
        return aggregate_neighbors(x, edge_index, edge_weight, self.get_mapper(), self.get_reducer(),
                                   self.get_updater())
