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
