# coding=utf-8
"""N>1 path on CPU: world_size-2 (and 3) gloo process groups drive tf_geometric_amd.dist.sharded with the numpy test
backend; the concatenated per-rank rows must equal the single-graph oracle."""

import numpy as np
import pytest

from conftest import assert_parity
import dist_worker


@pytest.mark.parametrize("world,skew,rounds", [(2, False, None), (2, True, 3), (3, True, 4), (2, False, 16), (8, False, None)])
def test_sharded_matches_oracle_gloo(tmp_path, world, skew, rounds):
    """rounds = number of all-to-all-v rounds the halo travels in (pipelined with the per-round reduce passes)."""
    port = dist_worker.free_port()
    parts = dist_worker.spawn(world, use_gpu=False, skew=skew, path=str(tmp_path), port=port, rounds=rounds)
    parts = dist_worker.check_against_reference(parts, skew, assert_parity)
    edges = [p["edges"] for p in parts]
    assert sum(edges) > 0 and all(p["n_halo"] > 0 for p in parts)
    if not skew:   # edge-balanced split: no shard is more than 25% off the mean
        assert max(edges) <= 1.25 * (sum(edges) / world)


@pytest.mark.parametrize("world,skew,rounds", [(2, True, 2), (3, False, None)])
def test_from_partitioned_matches_oracle_gloo(tmp_path, world, skew, rounds):
    """ShardedGraph.from_partitioned: every rank starts from its own stripe of the edge list (nothing edge-sized is
    replicated); after the degree all-reduce and the edge all-to-all-v the shards give the same layer outputs."""
    port = dist_worker.free_port()
    parts = dist_worker.spawn(world, use_gpu=False, skew=skew, path=str(tmp_path), port=port, rounds=rounds,
                              partitioned=True)
    parts = dist_worker.check_against_reference(parts, skew, assert_parity)
    assert sum(p["edges"] for p in parts) > 0


def test_from_partitioned_single_rank():
    res = {}
    dist_worker.run_checks(0, 1, use_gpu=False, skew=False, results=res, partitioned=True)
    dist_worker.check_against_reference([res[0]], False, assert_parity)


def test_single_rank_no_process_group():
    res = {}
    dist_worker.run_checks(0, 1, use_gpu=False, skew=True, results=res)
    dist_worker.check_against_reference([res[0]], True, assert_parity)
    assert res[0]["n_halo"] == 0


def test_edge_balanced_bounds():
    from tf_geometric_amd.dist.sharded import edge_balanced_bounds
    rp = np.array([0, 10, 10, 10, 40, 41, 42, 100], dtype=np.int64)
    b = edge_balanced_bounds(rp, 4)
    assert b[0] == 0 and b[-1] == 7 and (np.diff(b) >= 0).all()
    assert list(edge_balanced_bounds(np.zeros(9, np.int64), 4)) == [0, 2, 4, 6, 8]
    assert list(edge_balanced_bounds(rp, 1)) == [0, 7]


@pytest.mark.parametrize("world,skew,rounds,hub", [(2, False, 2, None), (3, True, 3, None), (1, True, None, None),
                                                   (2, True, 2, 8), (8, False, 4, None)])
def test_sharded_training_gradients_gloo(tmp_path, world, skew, rounds, hub):
    """Sharded backward: reverse halo all-to-all-v, owner-side accumulate in fixed peer order, weight-gradient
    all-reduce.  d/dx (per owner), d/dkernel and d/dbias (summed over ranks) must equal single-process float64 autograd
    over the oracle's normalised adjacency — what tf.GradientTape produces in the reference's training loops
    (demo/demo_gcn.py:68-77; distributed: demo/demo_distributed_gcn.py:52-57,99)."""
    if world == 1:
        parts = [dist_worker.run_training(0, 1, False, skew)]
    else:
        port = dist_worker.free_port()
        parts = dist_worker.spawn_training(world, False, skew, str(tmp_path), port, rounds=rounds, hub_threshold=hub)
    ref = dist_worker.training_reference(skew)
    parts = sorted(parts, key=lambda p: p["lo"])
    assert_parity(np.concatenate([p["out"] for p in parts]), ref["out"], what="sharded trainable forward")
    assert_parity(np.concatenate([p["dx"] for p in parts]), ref["dx"], tol=2e-5, what="sharded d/dx")
    assert_parity(np.concatenate([p["dx_mean"] for p in parts]), ref["dx_mean"], tol=2e-5, what="sharded mean d/dx")
    for p in parts:                       # every rank holds the SAME all-reduced weight gradients
        assert_parity(p["dk"], ref["dk"], tol=1e-4, what="all-reduced d/dkernel")
        assert_parity(p["db"], ref["db"], tol=1e-4, what="all-reduced d/dbias")
    dist_worker.check_training_extras(parts, ref, assert_parity)     # max / GAT / max-pool SAGE through halo_table
    for p in parts:                       # static layer-0 features: same rows, and NO exchange after the preparation
        assert np.array_equal(p["static_sum"], p["static_sum_ref"]) and np.array_equal(p["static_mean"], p["static_mean_ref"])
        assert p["static_exchanges"] == 0
    if world > 1:                         # ... which are the sum of different local parts
        assert np.abs(parts[0]["dk_local"] - parts[1]["dk_local"]).max() > 1e-3
        assert_parity(sum(p["dk_local"] for p in parts), ref["dk"], tol=1e-4, what="sum of local d/dkernel")


def test_column_chunked_halo_bounds_the_table_gloo(tmp_path):
    """aggregate_chunked(num_splits) (the reference's num_splits, utils/tf_sparse_utils.py:71-90): same rows as the
    unchunked pass, with a source table num_splits times smaller."""
    port = dist_worker.free_port()
    parts = dist_worker.spawn_training(2, False, True, str(tmp_path), port, rounds=2, num_splits=4)   # F = 12 -> four 3-column chunks
    for p in parts:
        assert np.array_equal(p["chunked"], p["whole"])
        assert p["chunk_table_floats"] * 4 == p["full_table_floats"]


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_long_spans_are_chunked_gloo(tmp_path, world):
    """A forced low hub threshold makes spans "long" on the small test graph: the reduce passes take their chunk lists
    and the sharded GAT merges per-row PART LISTS (whole passes + chunks of long spans) instead of two fixed states —
    same rows as the oracle."""
    if world == 1:
        res = {}
        dist_worker.run_checks(0, 1, use_gpu=False, skew=True, results=res, hub_threshold=8)
        parts = [res[0]]
    else:
        port = dist_worker.free_port()
        parts = dist_worker.spawn(world, use_gpu=False, skew=True, path=str(tmp_path), port=port, rounds=2,
                                  hub_threshold=8)
    assert all(p["gat_used_parts"] for p in parts)
    dist_worker.check_against_reference(parts, True, assert_parity)


@pytest.mark.parametrize("pct,expect_dense", [("60", True), ("0", False)])
def test_dense_peers_skip_the_pack_gloo(tmp_path, pct, expect_dense):
    """A peer whose block is referenced to >= TFGX_DENSE_PEER_PCT percent is requested WHOLE (its owner sends the block
    without packing; uniform random graphs at 8 GPUs: every peer).  Same rows as the oracle either way; with the rule on
    the small uniform test graph takes it (nothing is packed), with it off everything is packed."""
    port = dist_worker.free_port()
    parts = dist_worker.spawn(2, use_gpu=False, skew=False, path=str(tmp_path), port=port, rounds=3,
                              env={"TFGX_DENSE_PEER_PCT": pct})
    parts = dist_worker.check_against_reference(parts, False, assert_parity)
    for p in parts:
        assert p["rows_sent"] > 0
        if expect_dense:
            assert p["dense_send"] == 1 and p["rows_packed"] == 0
        else:       # rule off: a block still goes unpacked when a peer happens to reference every row of it
            assert p["rows_packed"] == p["rows_sent"] - p["dense_send"] * (p["hi"] - p["lo"])
    tr = dist_worker.spawn_training(2, False, False, str(tmp_path), dist_worker.free_port(), rounds=3, env={"TFGX_DENSE_PEER_PCT": pct})
    ref = dist_worker.training_reference(False)
    tr = sorted(tr, key=lambda p: p["lo"])
    assert_parity(np.concatenate([p["dx"] for p in tr]), ref["dx"], tol=2e-5, what="sharded d/dx, dense peers " + pct)
    dist_worker.check_training_extras(tr, ref, assert_parity)


@pytest.mark.parametrize("world", [1, 2])
def test_self_halo_mode_moves_own_rows_through_the_exchange(tmp_path, world):
    """self_halo_rows (the test mode the world-size-1 RCCL run on the GPU box relies on): only a third of a rank's rows
    are resident sources, the rest arrive through the halo exchange from the rank ITSELF — forward layers and the
    reverse exchange of the training path give the same rows as the oracle."""
    if world == 1:
        res = {}
        dist_worker.run_checks(0, 1, use_gpu=False, skew=True, results=res, rounds=2, self_halo=True)
        parts = [res[0]]
        tr = [dist_worker.run_training(0, 1, False, True, rounds=2, self_halo=True)]
    else:
        port = dist_worker.free_port()
        parts = dist_worker.spawn(world, use_gpu=False, skew=True, path=str(tmp_path), port=port, rounds=2, self_halo=True)
        tr = dist_worker.spawn_training(world, False, True, str(tmp_path), dist_worker.free_port(), rounds=2, self_halo=True)
    parts = dist_worker.check_against_reference(parts, True, assert_parity)
    assert all(p["n_halo"] > 0 and p["rows_sent"] > 0 for p in parts)
    ref = dist_worker.training_reference(True)
    tr = sorted(tr, key=lambda p: p["lo"])
    assert_parity(np.concatenate([p["out"] for p in tr]), ref["out"], what="self-halo trainable forward")
    assert_parity(np.concatenate([p["dx"] for p in tr]), ref["dx"], tol=2e-5, what="self-halo d/dx")
    assert_parity(np.concatenate([p["dx_mean"] for p in tr]), ref["dx_mean"], tol=2e-5, what="self-halo mean d/dx")
    dist_worker.check_training_extras(tr, ref, assert_parity)


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_strict_transport_failure_is_agreed(tmp_path, fail_rank):
    """transport="tfgx_dist" (what bench.py --gpus N asks for by name) never degrades and never hangs: the ranks agree on
    their local preconditions over the control channel BEFORE any RCCL call, so a failure on one rank (fail_rank = 1: its
    library load is made to fail) or on all of them (this box has no GPU) raises TfgxDistUnavailable on every rank."""
    import torch
    if fail_rank is None and torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (every rank then fails require_gpu)")
    port = dist_worker.free_port()
    msgs = dist_worker.spawn_agree(2, str(tmp_path), port, fail_rank)
    assert all("failed on" in m and "of 2 ranks" in m for m in msgs), msgs
    if fail_rank is not None:
        assert all("1 of 2 ranks" in m and "rank 1: TfgxError: injected" in m for m in msgs), msgs
    else:
        assert all("2 of 2 ranks" in m for m in msgs), msgs
