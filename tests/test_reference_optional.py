# coding=utf-8
"""Runs ONLY where the real reference is importable (tensorflow + tf_sparse + /root/reference): asserts
oracle == tf_geometric on the golden inputs, which would close the "parity unpinned" gap.
In this image neither package exists and there is no network, so these tests skip."""
import os
import sys

import numpy as np
import pytest

from conftest import assert_parity, ROOT

tf = pytest.importorskip("tensorflow", reason="TensorFlow is not installed in this image")
pytest.importorskip("tf_sparse", reason="tf_sparse is not installed in this image")
if not os.path.isdir("/root/reference/tf_geometric"):
    pytest.skip("reference checkout not present", allow_module_level=True)
sys.path.insert(0, "/root/reference")


def test_oracle_equals_reference_on_golden_inputs(oracle):
    import tf_geometric as tfg_ref
    g = np.load(os.path.join(ROOT, "tests", "golden", "hot_path_small.npz"))
    x, ei, w = g["x"], g["edge_index"], g["edge_weight"]
    ref = tfg_ref.nn.aggregate_neighbors(x, ei, w, tfg_ref.nn.gcn_mapper, tfg_ref.nn.sum_reducer,
                                         tfg_ref.nn.identity_updater).numpy()
    assert_parity(oracle.aggregate_neighbors(x, ei, w, oracle.gcn_mapper, oracle.sum_reducer,
                                             oracle.identity_updater), ref, what="oracle vs reference")
    ref_max = tfg_ref.nn.aggregate_neighbors(x, ei, w, tfg_ref.nn.gcn_mapper, tfg_ref.nn.max_reducer,
                                             tfg_ref.nn.identity_updater).numpy()
    assert np.array_equal(g["max_out"], ref_max)
    import tf_sparse as tfs
    adj = tfs.SparseMatrix(ei, w, [x.shape[0], x.shape[0]])
    ref_gcn = tfg_ref.nn.gcn(x, adj, g["gcn_kernel"], g["gcn_bias"], activation=tf.nn.relu).numpy()
    assert_parity(g["gcn_out"], ref_gcn, what="golden gcn vs reference")
    ref_gat = tfg_ref.nn.gat(x, ei, g["gat_wq"], g["gat_bq"], tf.nn.relu, g["gat_wk"], g["gat_bk"], tf.nn.relu,
                             g["gat_wv"], g["gat_b"], tf.nn.relu, num_heads=4).numpy()
    assert_parity(g["gat_out"], ref_gat, what="golden gat vs reference")
