# coding=utf-8
"""The drop-in boundary without Python in the loop: examples/c_abi_demo.cpp is a plain HIP host program that includes
include/tfgx.h, links libtfgx.so, builds a plan, runs the weighted segment-sum (+ implicit self-loops) and the MFMA
GEMM on device buffers it allocated itself with hipMalloc, and checks both against scalar loops."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_host_program_through_the_c_abi(tfg):
    from tf_geometric_amd import _build
    demo = _build.DEMO_BIN
    if not os.path.exists(demo):
        _build.build_c_abi_demo(verbose=False)
    res = subprocess.run([demo], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode()
    assert res.returncode == 0 and "C_ABI_DEMO_OK" in text, text


def test_c_host_halo_exchange_through_rccl(tfg):
    """examples/c_abi_halo_demo.cpp: tfgx_halo_plan_create / exchange_start / exchange_finish (include/tfgx_dist.h) from
    a torch-free C++ host with its own ncclComm_t — two rounds of packed rows really travel through RCCL (1-rank
    communicator, the rank requests rows from itself; a multi-rank exchange needs several GPUs)."""
    from tf_geometric_amd import _build
    demo = _build.HALO_DEMO_BIN
    if not os.path.exists(demo):
        _build.build_dist(verbose=False)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(["timeout", "120", demo], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    text = res.stdout.decode()
    assert res.returncode == 0 and "c_abi_halo_demo: OK" in text, text
