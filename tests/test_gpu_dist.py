# coding=utf-8
"""Sharded path with the HIP backend on the GPU box: world_size 1 in-process, and world_size 2 as two processes that
share cuda:0 with a gloo group (rows staged through the host) — the kernels and the plan code are the product's,
only the transport differs from the RCCL path the 8-GPU bench uses."""
import random

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
    port = 31500 + random.randint(0, 2000)
    parts = dist_worker.spawn(2, use_gpu=True, skew=skew, path=str(tmp_path), port=port, rounds=rounds)
    dist_worker.check_against_reference(parts, skew, assert_parity)


def test_from_partitioned_world2_hip_gloo_transport(tfg, tmp_path):
    port = 33600 + random.randint(0, 2000)
    parts = dist_worker.spawn(2, use_gpu=True, skew=True, path=str(tmp_path), port=port, rounds=3, partitioned=True)
    dist_worker.check_against_reference(parts, True, assert_parity)


def test_rccl_world1_api_smoke(tfg):
    """The RCCL calls of the sharded path, on a one-rank "nccl" group (tests/rccl_world1_smoke.py)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(35600 + random.randint(0, 2000)), RANK="0",
               WORLD_SIZE="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1_smoke.py")
    res = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode()
    assert res.returncode == 0 and "RCCL_WORLD1_OK" in text and "True" in text and "False" not in text, text
