# coding=utf-8
"""N>1 path on CPU: world_size-2 (and 3) gloo process groups drive tf_geometric_amd.dist.sharded with the numpy test
backend; the concatenated per-rank rows must equal the single-graph oracle."""
import random

import numpy as np
import pytest

from conftest import assert_parity
import dist_worker


@pytest.mark.parametrize("world,skew,rounds", [(2, False, None), (2, True, 3), (3, True, 4), (2, False, 16)])
def test_sharded_matches_oracle_gloo(tmp_path, world, skew, rounds):
    """rounds = number of all-to-all-v rounds the halo travels in (pipelined with the per-round reduce passes)."""
    port = 29500 + random.randint(0, 2000)
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
    port = 29500 + random.randint(2001, 4000)
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
