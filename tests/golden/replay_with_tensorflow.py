# coding=utf-8
"""Pin the oracle against the REAL reference: replay tests/golden/hot_path_small.npz through tf_geometric itself.

Cannot run in the build image (tensorflow and tf_sparse are absent and there is no network — SURVEY.md §8c); run it on
any box with `pip install tensorflow tf_sparse tf_geometric` (CPU is fine):

    python tests/golden/replay_with_tensorflow.py [path/to/hot_path_small.npz]

For every stored output it calls the reference's own functional API on the stored inputs (same argument order as
tf_geometric/nn/conv/gcn.py:225, gat.py:13, graph_sage.py:9, nn/kernel/map_reduce.py:45) and reports
max |reference - stored|.  All four lines inside 1e-5 + 1e-5*|ref| turn "parity unpinned" into "pinned" for the
fixture; a line outside it names the semantic the oracle restated wrongly (DESIGN.md §5 lists the assumed ones)."""
import os
import sys

import numpy as np


def main(path):
    import tensorflow as tf
    import tf_geometric as tfg
    from tf_sparse import SparseMatrix

    d = np.load(path)
    x, ei, w = d["x"], d["edge_index"], d["edge_weight"]
    n = x.shape[0]

    def report(name, ref, stored):
        ref = np.asarray(ref)
        err = np.abs(ref - stored) - 1e-5 * np.abs(stored)
        print("{:10s} max(|ref - stored| - 1e-5|stored|) = {:.3e}   {}".format(
            name, float(err.max()), "OK" if float(err.max()) <= 1e-5 else "OUTSIDE THE BAND"))
        return float(err.max()) <= 1e-5

    ok = True
    adj = SparseMatrix(ei, value=w, shape=[n, n])
    ok &= report("gcn", tfg.nn.gcn(x, adj, d["gcn_kernel"], d["gcn_bias"], activation=tf.nn.relu), d["gcn_out"])
    from tf_geometric.nn.conv.gcn import gcn_mapper
    ok &= report("max", tfg.nn.aggregate_neighbors(x, ei, w, gcn_mapper, tfg.nn.max_reducer, tfg.nn.identity_updater),
                 d["max_out"])
    ok &= report("mean_sage", tfg.nn.mean_graph_sage(x, ei, w, d["sage_self"], d["sage_neigh"], d["sage_bias"],
                                                     activation=tf.nn.relu, concat=True, normalize=True), d["sage_out"])
    ok &= report("gat", tfg.nn.gat(x, ei, d["gat_wq"], d["gat_bq"], tf.nn.relu, d["gat_wk"], d["gat_bk"], tf.nn.relu,
                                   d["gat_wv"], d["gat_b"], tf.nn.relu, num_heads=4), d["gat_out"])
    return 0 if ok else 1


if __name__ == "__main__":
    default = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hot_path_small.npz")
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else default))
