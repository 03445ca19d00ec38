# coding=utf-8
"""The C-ABI library loads (no GPU needed) and exports every symbol include/tfgx.h declares; the ctypes structs
match the header's structs; host-side argument checks fire; the product path refuses to run without a GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "tfgx.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfgx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tf_geometric_amd import _lib
    lib = _lib.load_library()
    names = _declared_functions()
    assert len(names) >= 19
    for name in names:
        assert hasattr(lib, name), "libtfgx.so does not export {}".format(name)
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree: {}".format(
        set(names) ^ set(_lib.SIGNATURES))
    assert lib.tfgx_version() == 114          # include/tfgx.h TFGX_ABI_VERSION == _lib.ABI_VERSION


def test_structs_match_header_layout(tmp_path):
    """Compile a tiny C program against include/tfgx.h and compare sizeof/offsetof with the ctypes mirrors."""
    from tf_geometric_amd import _lib
    src = tmp_path / "layout.c"
    fields_r = [f for f, _ in _lib.ReduceArgs._fields_]
    fields_g = [f for f, _ in _lib.GatArgs._fields_]
    body = ['#include <stdio.h>', '#include <stddef.h>', '#include "tfgx.h"', 'int main(void){',
            'printf("%zu\\n", sizeof(tfgx_reduce_args));']
    body += ['printf("%zu\\n", offsetof(tfgx_reduce_args, {}));'.format(f) for f in fields_r]
    body += ['printf("%zu\\n", sizeof(tfgx_gat_args));']
    body += ['printf("%zu\\n", offsetof(tfgx_gat_args, {}));'.format(f) for f in fields_g]
    body += ['return 0;}']
    src.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals[0] == ctypes.sizeof(_lib.ReduceArgs)
    for f, off in zip(fields_r, vals[1:1 + len(fields_r)]):
        assert getattr(_lib.ReduceArgs, f).offset == off, f
    rest = vals[1 + len(fields_r):]
    assert rest[0] == ctypes.sizeof(_lib.GatArgs)
    for f, off in zip(fields_g, rest[1:]):
        assert getattr(_lib.GatArgs, f).offset == off, f


def test_argument_validation_without_gpu():
    """Host-side checks return TFGX_ERR_INVALID_ARG before anything touches a device."""
    from tf_geometric_amd import _lib
    lib = _lib.load_library()
    assert lib.tfgx_segment_reduce_f32(None, None) == 1
    assert b"args is null" in lib.tfgx_last_error()
    a = _lib.ReduceArgs()
    a.n_dst, a.F, a.op = 4, 0, 0
    assert lib.tfgx_segment_reduce_f32(ctypes.byref(a), None) == 1
    a.F, a.op = 8, 7
    assert lib.tfgx_segment_reduce_f32(ctypes.byref(a), None) == 1 and b"bad op" in lib.tfgx_last_error()
    assert lib.tfgx_gemm_bias_act_f32(None, 4, None, 4, None, 0, None, 4, 2, 0, 4, None) == 1
    assert lib.tfgx_gcn_norm_edges_f32(None, None, None, 3, None, None, 9, 1.0, 1, 1, None, None, None) == 1
    assert lib.tfgx_csr_plan_workspace_bytes(10, 100) > 800
    assert lib.tfgx_build_csr_by_dst(None, None, -1, 3, 3, None, None, None, None, 0, None) == 1
    # entry points added later in the round: same contract (argument errors are reported before any device work)
    assert lib.tfgx_segment_topk(None, None, -1, 3, 1, 0.0, None, None, None, 0, None) == 1
    assert lib.tfgx_segment_topk(None, None, 5, 3, -1, -0.5, None, None, None, 0, None) == 1
    assert lib.tfgx_segment_topk(None, None, 5, 3, 1, 0.0, None, None, None, 0, None) == 1 and b"out_count" in lib.tfgx_last_error()
    assert lib.tfgx_segment_topk_workspace_bytes(1000, 10) > 1000 * 24 and lib.tfgx_segment_topk_workspace_bytes(-1, 1) == 0
    assert lib.tfgx_gemm_workspace_bytes(2708, 1433, 256) >= 2 * 4 * 2708 * 256      # small M, long K: split-K
    assert 0 < lib.tfgx_gemm_workspace_bytes(2400000, 100, 256) <= 4096               # plenty of tiles: no split, only the row kernel's tile counters
    assert lib.tfgx_gemm_workspace_bytes(170000, 128, 256) == 0                       # short launches keep the fixed tile map
    assert lib.tfgx_gemm_bias_act_cols_ws_f32(None, 4, None, 4, None, 0, 9, None, 4, 2, 4, 4, None, 0, None) == 1
    assert b"act_cols" in lib.tfgx_last_error()
    assert lib.tfgx_segment_max_with_count_f32(None, None, None, 4, None, 2, 8, None, 8, None, 8, None) == 1
    assert lib.tfgx_segment_max_backward_w_f32(None, None, None, 4, None, 8, 8, None, 8, None, 4, None, None) == 1
    # phases outside {1, 2, 3}; and a fused aggregate -> GEMM shape that does not fit LDS is refused on the host
    assert lib.tfgx_segment_max_backward_mask_phases_f32(None, None, None, 4, 8, None, 8, 8, None, 8, None, 8, None, 8, None, 8,
                                                         None, None, None, None, 4, None, 8, None, 0, 0, None) == 1
    assert lib.tfgx_aggregate_gemm_fits(128, 256) == 1 and lib.tfgx_aggregate_gemm_fits(100, 256) == 1
    assert lib.tfgx_aggregate_gemm_fits(102, 16) == 0 and lib.tfgx_aggregate_gemm_fits(100, 257) == 0
    g = _lib.GatArgs()
    g.H, g.d, g.dv, g.n_dst, g.scale, g.drop_rate = 2, 4, 4, 3, 2.0, 1.5
    assert lib.tfgx_gat_fused_f32(ctypes.byref(g), None) == 1
    assert lib.tfgx_dropout_keep(7, 3, 0.0) == 1                                      # rate 0 keeps everything
    kept = sum(lib.tfgx_dropout_keep(0x1234567890, i, 0.25) for i in range(4000))
    assert 2850 < kept < 3150                                                         # ~75 %
    assert [lib.tfgx_dropout_keep(99, i, 0.5) for i in range(64)] == [lib.tfgx_dropout_keep(99, i, 0.5) for i in range(64)]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_gpu():
    import tf_geometric_amd as tfg
    x = np.ones((4, 3), np.float32)
    ei = np.array([[0, 1], [1, 0]], np.int32)
    with pytest.raises(tfg._lib.TfgxError):
        tfg.nn.aggregate_neighbors(x, ei, None)
    with pytest.raises(tfg._lib.TfgxError):
        tfg.layers.GCN(4)([x, ei])
    with pytest.raises(tfg._lib.TfgxError):
        tfg.layers.GAT(4)([x, ei])
    with pytest.raises(tfg._lib.TfgxError):
        tfg.layers.MeanGraphSage(4)([x, ei])
    with pytest.raises(tfg._lib.TfgxError):
        tfg.SparseMatrix(ei, None, [4, 4]) @ x


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "tf_geometric_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith((".py", ".hip", ".h")):
                continue
            for line in open(os.path.join(dirpath, fn)).read().splitlines():
                assert not re.match(r"\s*(from|import)\s+oracle", line), "{} imports oracle".format(fn)
                assert "libtfg_oracle" not in line and "tfg_oracle" not in line, "{} references the oracle".format(fn)


def test_activation_resolution_and_layer_contract():
    import tf_geometric_amd as tfg
    from tf_geometric_amd.activations import resolve
    assert resolve(None) == (0, None) and resolve("relu") == (1, None) and resolve(tfg.relu) == (1, None)
    assert resolve(torch.relu) == (1, None)
    code, post = resolve(torch.tanh)
    assert code == 0 and post is torch.tanh
    with pytest.raises(ValueError):
        resolve("gelu")
    with pytest.raises(Exception):
        tfg.layers.GCN(4, num_splits=2, num_or_size_splits=2)                 # layers/conv/gcn.py:82-83
    with pytest.raises(Exception):
        tfg.layers.MeanGraphSage(5, concat=True)                              # layers/conv/graph_sage.py:36-37
    from tf_geometric_amd.nn.conv.gcn import compute_cache_key
    assert compute_cache_key("both", True, True, True, False) == "gcn_normed_adj_both_True_True_True_False"


def test_synthetic_inputs_are_reproducible():
    from tf_geometric_amd import synthetic
    a = synthetic.synthetic_edges(1000, 5000, seed=0)
    b = synthetic.synthetic_edges(1000, 5000, seed=0)
    assert np.array_equal(a, b) and a.dtype == np.int32 and a.shape[0] == 2
    half = a.shape[1] // 2
    assert np.array_equal(a[0, :half], a[1, half:]) and np.array_equal(a[1, :half], a[0, half:])   # (a,b) then (b,a)
    assert (a[0] != a[1]).all()


def test_dist_library_exports_every_declared_symbol():
    """include/tfgx_dist.h (the halo-exchange C ABI: ncclComm_t + streams) vs lib/libtfgx_dist.so; host
    argument checks run without a GPU (no RCCL call is reached)."""
    from tf_geometric_amd import _build
    with open(os.path.join(ROOT, "include", "tfgx_dist.h")) as fh:
        src = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tfgx_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 19, names          # 19 = + tfgx_dist_comm_abort (round 5)
    if not os.path.exists(_build.DIST_LIB):
        _build.build_dist(verbose=False)
    lib = ctypes.CDLL(_build.DIST_LIB)
    for name in names:
        assert hasattr(lib, name), "libtfgx_dist.so does not export {}".format(name)
    lib.tfgx_dist_last_error.restype = ctypes.c_char_p
    plan = ctypes.c_void_p()
    cnt = (ctypes.c_int64 * 4)(0, 3, 0, 2)
    assert lib.tfgx_halo_plan_create(2, 5, 2, cnt, cnt, None, None, ctypes.byref(plan)) == 1      # rank outside world
    assert b"bad world" in lib.tfgx_dist_last_error()
    assert lib.tfgx_halo_plan_create(2, 0, 2, cnt, cnt, None, None, ctypes.byref(plan)) == 1      # rows to pack, no index list
    dense = (ctypes.c_int64 * 4)(-1, 0, -1, 3)          # every non-empty (round, peer) entry is a contiguous block:
    rc = lib.tfgx_halo_plan_create(2, 0, 2, cnt, cnt, dense, None, ctypes.byref(plan))            # nothing to pack: the
    assert rc in (0, 4)           # arguments are accepted; without a GPU the plan's hipEventCreate then fails (TFGX_ERR_HIP)
    if rc == 0:
        lib.tfgx_halo_plan_rows_packed.restype = lib.tfgx_halo_plan_rows_sent.restype = ctypes.c_int64
        assert lib.tfgx_halo_plan_rows_packed(plan) == 0 and lib.tfgx_halo_plan_rows_sent(plan) == 5
        assert lib.tfgx_halo_plan_destroy(plan) == 0
    else:
        assert b"hipEventCreate" in lib.tfgx_dist_last_error()
    assert lib.tfgx_dist_comm_init(0, 0, None, None) == 1 and lib.tfgx_dist_unique_id(None) == 1
    assert lib.tfgx_dist_comm_info(None, None, None, None) == 1
    assert lib.tfgx_alltoallv(None, None, None, None, 4, 2, None, None) == 1
    assert lib.tfgx_halo_exchange_finish(None, 0, None) == 1
    assert lib.tfgx_halo_reverse_start(None, None, 4, None, 0, None, None, None) == 1
    assert lib.tfgx_halo_reverse_start_round(None, 0, None, 4, None, 0, None, None, None) == 1
    assert lib.tfgx_halo_reverse_finish(None, None, 4, 4, None, None) == 1


def test_tf_shim_compiles_against_mock_headers():
    """integration/tf_shim/tfgx_tf_ops.cc (the tf.load_op_library binding of INTEGRATION.md) type-checks against
    include/tfgx.h and a mock of the TensorFlow headers it uses: every C-ABI call in it has the declared argument
    types.  TensorFlow itself is not installable here — the shim is a sketch that has never been linked or run."""
    shim = os.path.join(ROOT, "integration", "tf_shim")
    res = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(shim, "mock"),
                          "-I", os.path.join(ROOT, "include"), os.path.join(shim, "tfgx_tf_ops.cc")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert res.returncode == 0, res.stdout.decode()
    src = open(os.path.join(shim, "tfgx_tf_ops.cc")).read()
    ops = ("TfgxBuildCsrByDst", "TfgxSegmentReduce", "TfgxGatFused", "TfgxGcnNormEdges", "TfgxGemmBiasAct", "TfgxAggregateGemm",
           # round 3: the backward ops and the sharded path
           "TfgxSddmm", "TfgxPermuteRows", "TfgxGemmTn", "TfgxReluBackward", "TfgxHaloExchange", "TfgxHaloReverse",
           "TfgxAllReduceSum")
    for op in ops:
        assert 'REGISTER_OP("{}")'.format(op) in src and 'Name("{}")'.format(op) in src
    # the Python side (gradient registration, GCN layer over the ops): compiles, and names only ops the library registers
    import py_compile
    pyfile = os.path.join(shim, "tfgx_tf.py")
    py_compile.compile(pyfile, doraise=True)
    pysrc = open(pyfile).read()
    snake = {re.sub(r"(?<!^)(?=[A-Z])", "_", op).lower() for op in ops}
    used = set(re.findall(r"ops\.(tfgx_[a-z_]+)\(", pysrc))
    assert used and used <= snake, used - snake
    for grad in ("TfgxSegmentReduce", "TfgxGemmBiasAct", "TfgxHaloExchange"):
        assert '@tf.RegisterGradient("{}")'.format(grad) in pysrc


def test_host_code_under_address_sanitizer():
    """SURVEY.md §5 (memory-error detection): the host side of the C-ABI library built with AddressSanitizer
    (lib/asan/libtfgx.so, device code untouched) survives a sweep over its entry points' validation / query paths —
    including a deliberately short output buffer — without an ASAN report."""
    from tf_geometric_amd import _build
    rt = _build.asan_runtime()
    if rt is None:
        pytest.skip("clang's shared ASAN runtime is not in this ROCm image")
    lib = _build.build_asan(verbose=False)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    res = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "asan_host_check.py"), lib], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = res.stdout.decode()
    assert res.returncode == 0 and "ASAN_HOST_CHECK_OK" in text and "AddressSanitizer" not in text, text[-3000:]
