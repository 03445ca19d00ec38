# coding=utf-8
"""Sharded path with the HIP backend on the GPU box: world_size 1 in-process, and world_size 2 as two processes that
share cuda:0 with a gloo group (rows staged through the host) — the kernels and the plan code are the product's,
only the transport differs from the RCCL path the 8-GPU bench uses."""

import pytest

from conftest import assert_parity
import dist_worker

pytestmark = pytest.mark.gpu


def test_sharded_world1_hip(tfg):
    res = {}
    dist_worker.run_checks(0, 1, use_gpu=True, skew=True, results=res)
    dist_worker.check_against_reference([res[0]], True, assert_parity)


@pytest.mark.parametrize("skew,rounds", [(False, None), (True, 4)])
def test_sharded_world2_hip_gloo_transport(tfg, tmp_path, skew, rounds):
    port = dist_worker.free_port()
    parts = dist_worker.spawn(2, use_gpu=True, skew=skew, path=str(tmp_path), port=port, rounds=rounds)
    dist_worker.check_against_reference(parts, skew, assert_parity)


def test_from_partitioned_world2_hip_gloo_transport(tfg, tmp_path):
    port = dist_worker.free_port()
    parts = dist_worker.spawn(2, use_gpu=True, skew=True, path=str(tmp_path), port=port, rounds=3, partitioned=True)
    dist_worker.check_against_reference(parts, True, assert_parity)


def test_rccl_world1_api_smoke(tfg):
    """The RCCL calls of the sharded path, on a one-rank "nccl" group (tests/rccl_world1_smoke.py)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(dist_worker.free_port()), RANK="0",
               WORLD_SIZE="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1_smoke.py")
    res = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode()
    assert res.returncode == 0 and "RCCL_WORLD1_OK" in text and "True" in text and "False" not in text, text


@pytest.mark.parametrize("partitioned", [False, True])
def test_sharded_layers_through_the_c_abi_exchange_world1(tfg, partitioned):
    """The product transport (libtfgx_dist.so: in-process ncclComm_t, grouped ncclSend / ncclRecv on the second HIP stream)
    carrying real rows on ONE GPU: self-halo test mode, no torch.distributed group at all."""
    res = {}
    dist_worker.run_checks(0, 1, use_gpu=True, skew=True, results=res, rounds=3, partitioned=partitioned, self_halo=True)
    p = res[0]
    assert p["transport"] == "tfgx_dist" and p["n_halo"] > 0 and p["rows_sent"] > 0 and p["rows_packed"] == p["rows_sent"]
    dist_worker.check_against_reference([p], True, assert_parity)


@pytest.mark.parametrize("skew", [True, False])
def test_sharded_training_through_the_c_abi_exchange_world1(tfg, skew):
    """Reverse exchange (tfgx_halo_reverse_start overlapped with the own-row part of the transposed pass, then
    tfgx_halo_reverse_finish) and the column-chunked halo (two exchanges in flight) through the product transport."""
    import numpy as np
    tr = dist_worker.run_training(0, 1, True, skew, rounds=3, num_splits=4, self_halo=True)
    ref = dist_worker.training_reference(skew)
    assert_parity(tr["out"], ref["out"], what="trainable forward (tfgx_dist)")
    assert_parity(tr["dx"], ref["dx"], tol=2e-5, what="d/dx (tfgx_dist reverse exchange)")
    assert_parity(tr["dx_mean"], ref["dx_mean"], tol=2e-5, what="mean d/dx (tfgx_dist)")
    assert_parity(tr["dk"], ref["dk"], tol=1e-4, what="d/dkernel")
    dist_worker.check_training_extras([tr], ref, assert_parity)
    assert np.array_equal(tr["chunked"], tr["whole"])
    # trainable max / max-pool SAGE / GAT: forwards ran span by span under the exchange (tracked max merged in the kernel
    # epilogue, GAT states merged with the softmax statistics written for the backward)
    # (the skewed graph has hub rows: those are chunked and do not track, so its max forward waits for the exchange)
    assert tr["counters"].get("gat_span_training_forwards", 0) >= 1, tr["counters"]
    assert skew or tr["counters"].get("gat_halo_first_backwards", 0) >= 1, tr["counters"]   # (hub sources: one pass)
    assert skew or tr["counters"].get("max_span_forwards", 0) >= 2, tr["counters"]       # max at 36 columns + max-pool SAGE
    assert skew or tr["counters"].get("max_halo_first_backwards", 0) >= 2, tr["counters"]   # ... and their halo-first backward


def test_tfgx_dist_world1_under_an_nccl_process_group(tfg):
    """The same, in a subprocess that first initialises a one-rank "nccl" torch.distributed group (the configuration
    bench.py --gpus N runs in: torch's group is the control channel, the halo rows travel through tfgx_dist's own
    communicator) — tests/tfgx_dist_world1.py."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(dist_worker.free_port()), RANK="0",
               WORLD_SIZE="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tfgx_dist_world1.py")
    res = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = res.stdout.decode()
    assert res.returncode == 0 and "TFGX_DIST_WORLD1_OK" in text, text[-3000:]


@pytest.mark.parametrize("world,skew,hub", [(1, True, None), (2, True, None), (2, False, None), (2, True, 8)])
def test_sharded_training_hip(tfg, tmp_path, world, skew, hub):
    """Sharded backward on the HIP kernels (transposed local pass, tfgx_scatter_add_rows_f32 owner-side accumulate,
    MFMA weight gradients) + the reverse exchange and weight-gradient all-reduce over a gloo group sharing cuda:0, and
    the column-chunked halo (bit-identical rows, a quarter of the table)."""
    import numpy as np
    if world == 1:
        parts = [dist_worker.run_training(0, 1, True, skew, num_splits=4, hub_threshold=hub)]
    else:
        port = dist_worker.free_port()
        parts = dist_worker.spawn_training(2, True, skew, str(tmp_path), port, rounds=3, num_splits=4, hub_threshold=hub)
    ref = dist_worker.training_reference(skew)
    parts = sorted(parts, key=lambda p: p["lo"])
    assert_parity(np.concatenate([p["out"] for p in parts]), ref["out"], what="sharded trainable forward (HIP)")
    assert_parity(np.concatenate([p["dx"] for p in parts]), ref["dx"], tol=2e-5, what="sharded d/dx (HIP)")
    assert_parity(np.concatenate([p["dx_mean"] for p in parts]), ref["dx_mean"], tol=2e-5, what="sharded mean d/dx (HIP)")
    # max aggregation, the GAT layer and (unskewed graph) the max-pool SAGE layer: differentiable halo table + the
    # single-GPU backward kernels on the shard's rectangular plan, halo-row gradients returned by the reverse exchange
    dist_worker.check_training_extras(parts, ref, assert_parity)
    for p in parts:
        assert_parity(p["dk"], ref["dk"], tol=1e-4, what="all-reduced d/dkernel (HIP)")
        assert_parity(p["db"], ref["db"], tol=1e-4, what="all-reduced d/dbias (HIP)")
        assert np.array_equal(p["chunked"], p["whole"])
        assert p["chunk_table_floats"] * 4 == p["full_table_floats"]
        assert np.array_equal(p["static_sum"], p["static_sum_ref"]) and np.array_equal(p["static_mean"], p["static_mean_ref"])
        assert p["static_exchanges"] == 0


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_long_spans_are_chunked_hip(tfg, tmp_path, world):
    """Forced low hub threshold on the HIP backend: chunked reduce passes, and the sharded GAT through raw-state launches
    over parts (tfgx_gat_fused_f32 with part rows / skipped long spans) + tfgx_gat_merge_parts_f32."""
    if world == 1:
        res = {}
        dist_worker.run_checks(0, 1, use_gpu=True, skew=True, results=res, hub_threshold=8)
        parts = [res[0]]
    else:
        port = dist_worker.free_port()
        parts = dist_worker.spawn(2, use_gpu=True, skew=True, path=str(tmp_path), port=port, rounds=2, hub_threshold=8)
    assert all(p["gat_used_parts"] for p in parts)
    dist_worker.check_against_reference(parts, True, assert_parity)


@pytest.mark.parametrize("self_halo", ["0", "1"])
def test_demo_sharded_gcn_trains(tfg, self_halo):
    """examples/demo_sharded_gcn.py (counterpart of the reference's demo/demo_distributed_gcn.py: there the graph is
    replicated and the gradients all-reduced, :52-57,99 — here the graph is sharded by destination range): the 2-layer GCN
    trains to far above 1/16 chance on one GPU, plain and with TFGX_DEMO_SELF_HALO=1 (three quarters of the source rows
    really travel through the RCCL exchange, forward and reverse, every step)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(dist_worker.free_port()), RANK="0",
               WORLD_SIZE="1", TFGX_DEMO_SELF_HALO=self_halo)
    res = subprocess.run([sys.executable, os.path.join(root, "examples", "demo_sharded_gcn.py"), "--steps", "40",
                          "--nodes", "30000", "--edges", "600000"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=600)
    text = res.stdout.decode()
    assert res.returncode == 0, text[-3000:]
    last = [ln for ln in text.splitlines() if ln.startswith("step = 40")][-1]
    acc = float(last.split("test accuracy = ")[1].split()[0])
    assert acc > 0.5, last
    assert ("transport tfgx_dist" in last) if self_halo == "1" else True


def test_sharded_span_passes_gather_one_burst_per_row(tfg):
    """plan.segment_reduce switched explicit-span launches (a row's own-source edges, then one sub-span per halo
    round — a handful of edges per row and pass) to one burst per gathered row, but the sharded backend filled its own
    ReduceArgs and left the kernel's 64-column blocks on: the row start-up was paid once per block.  ONE policy now
    (plan.wide_blocks_hint) — the dispatcher's choice for the same wide rows, whole rows vs class sub-spans."""
    import torch
    from tf_geometric_amd import _lib as L, synthetic
    from tf_geometric_amd.dist.sharded import HipBackend
    from tf_geometric_amd.plan import CsrPlan, wide_blocks_hint
    n, e, F = 40000, 3200000, 256
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=2))
    plan = CsrPlan.build(ei, n, n)
    be = HipBackend()
    x = torch.randn(n, F + 32, device="cuda")[:, :F]          # line-aligned rows, not a power-of-two stride
    out = torch.empty(n, F, device="cuda")
    whole = be.segment_reduce(plan.row_ptr, plan.row_ptr[1:], 1, plan.col, None, n, x, out, L.SUM, describe=True)
    # two classes: [row_ptr_k[2 r], row_ptr_k[2 r + 1]) own-source edges, [.. + 1, .. + 2) halo edges (here: split in the middle)
    mid = (plan.row_ptr[:-1] + plan.row_ptr[1:]) // 2
    rpk = torch.stack([plan.row_ptr[:-1], mid], dim=1).reshape(-1)
    rpk = torch.cat([rpk, plan.row_ptr[-1:]]).contiguous()
    spans = be.segment_reduce(rpk, rpk[1:], 2, plan.col, None, n, x, out, L.SUM, describe=True)
    assert whole.endswith(", 16>") and spans.endswith(", 0>"), (whole, spans)
    assert wide_blocks_hint(True, False, F + 32, e, n) == -1 and wide_blocks_hint(False, False, F + 32, e, n) == 0
    assert wide_blocks_hint(False, False, F + 32, 14 * n, n) == -1 and wide_blocks_hint(False, True, F + 32, e, n) == -1
    # and the numbers agree: the two class passes accumulated == the whole-row pass
    ref = be.segment_reduce(plan.row_ptr, plan.row_ptr[1:], 1, plan.col, None, n, x, torch.empty_like(out), L.SUM)
    got = be.segment_reduce(rpk, rpk[1:], 2, plan.col, None, n, x, torch.empty_like(out), L.SUM)
    got = be.segment_reduce(rpk[1:], rpk[2:], 2, plan.col, None, n, x, got, L.SUM, accumulate=True)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4)
