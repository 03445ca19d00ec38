# coding=utf-8
"""tf_geometric_amd — MI355X (gfx950) message-passing backend behind tf_geometric's own API.

    import tf_geometric_amd as tfg
    h = tfg.layers.GCN(16, activation=tfg.relu)([x, edge_index, edge_weight], cache=graph_cache)

Only the hot path of the reference is here: aggregate_neighbors / GCN / GAT / GraphSAGE,
the dense x @ W next to it, and dst-range sharding with halo exchange.  Compute = hand-written HIP kernels in
lib/libtfgx.so (C ABI: include/tfgx.h); importing works without a GPU, calling an operator does not.
"""
from . import _lib
from .activations import relu
from .plan import CsrPlan, prepare_static_features, release_static_features
from .sparse import SparseMatrix
from . import nn
from . import layers
from . import dist
from . import utils
from .graph_capture import CapturedForward, CapturedTrainStep

__version__ = "0.1.0"
