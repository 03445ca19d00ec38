# coding=utf-8
"""Layer API — same class names, constructor arguments and call signature as tf_geometric.layers."""
from .conv.gcn import GCN
from .conv.gat import GAT
from .conv.graph_sage import MeanGraphSage, SumGraphSage, GCNGraphSage, MeanPoolGraphSage, MaxPoolGraphSage
from .kernel.map_reduce import MapReduceGNN
from .conv.propagation import GIN, SGC, TAGCN, APPNP, SSGC, ChebyNet, LEConv
from .pool import CommonPool, MeanPool, SumPool, MaxPool, MinPool
