# coding=utf-8
"""Drives the HOST side of every C-ABI entry point of an AddressSanitizer build of libtfgx.so (no GPU needed: argument
validation, workspace queries, the dispatch description, the dropout hash, error strings).  Run by
tests/test_abi.py::test_host_code_under_address_sanitizer in a subprocess with the ASAN runtime preloaded:

    LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_check.py <libtfgx.so>

Any heap/stack/global overflow or use-after-free in the host code aborts the process with an ASAN report."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("tfgx_lib_table", os.path.join(ROOT, "tf_geometric_amd", "_lib.py"))
    # _lib imports torch only for the device helpers; the signature table is what is needed here
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = ctypes.CDLL(path)
    for name, (res, args) in mod.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    checked = 0
    assert lib.tfgx_version() >= 100
    a = mod.ReduceArgs()
    for n_dst, F, op in [(4, 0, 0), (4, 8, 7), (-1, 8, 0), (4, 8, 0)]:
        a.n_dst, a.F, a.op = n_dst, F, op
        assert lib.tfgx_segment_reduce_f32(ctypes.byref(a), None) == 1
        assert len(lib.tfgx_last_error()) > 0
        checked += 1
    buf = ctypes.create_string_buffer(8)                       # deliberately short: snprintf must truncate, not overflow
    a.n_dst, a.F, a.op, a.ldx, a.ldo = 4, 100, 0, 100, 100
    assert lib.tfgx_segment_reduce_describe(ctypes.byref(a), buf, 8) == 0 and len(buf.value) == 7
    big = ctypes.create_string_buffer(200)
    assert lib.tfgx_segment_reduce_describe(ctypes.byref(a), big, 200) == 0 and b"seg_reduce_kernel<" in big.value
    assert lib.tfgx_segment_reduce_describe(None, big, 200) == 1
    checked += 3
    assert lib.tfgx_gemm_bias_act_f32(None, 4, None, 4, None, 0, None, 4, 2, 0, 4, None) == 1
    assert lib.tfgx_gemm_bias_act_cols_ws_f32(None, 4, None, 4, None, 0, 9, None, 4, 2, 4, 4, None, 0, None) == 1
    assert lib.tfgx_gemm_workspace_bytes(2708, 1433, 256) > 0 and lib.tfgx_gemm_workspace_bytes(0, 1, 1) == 0
    assert lib.tfgx_gemm_tn_workspace_bytes(2400000, 100, 256, 1) > 0 and lib.tfgx_gemm_tn_workspace_bytes(5, 3000, 4, 0) > 0 and lib.tfgx_gemm_tn_workspace_bytes(0, 3, 4, 0) == 0
    assert lib.tfgx_gemm_tn_f32(None, 4, None, 4, 10, 4, 4, None, 4, None, None, 0, None) == 1
    assert lib.tfgx_transpose_f32(None, 1, 4, 4, None, 4, None) == 1
    assert lib.tfgx_gcn_norm_edges_f32(None, None, None, 3, None, None, 9, 1.0, 1, 1, None, None, None) == 1
    assert lib.tfgx_build_csr_by_dst(None, None, -1, 3, 3, None, None, None, None, 0, None) == 1
    assert lib.tfgx_csr_plan_workspace_bytes(10, 100) > 0
    assert lib.tfgx_segment_topk(None, None, 5, 3, 1, 0.0, None, None, None, 0, None) == 1
    assert lib.tfgx_segment_topk_workspace_bytes(1000, 10) > 0
    assert lib.tfgx_segment_max_with_arg_f32(None, None, None, 4, None, 8, 8, None, 8, None, 8, None, 8, None) == 1
    assert lib.tfgx_segment_max_backward_push_f32(None, None, None, 4, 0, None, 8, 8, None, 8, None, 8, None, 8, None, 8,
                                                  None, 8, None) == 1
    assert lib.tfgx_scatter_add_rows_f32(None, 4, None, 3, 4, None, 4, None) == 1
    assert lib.tfgx_sample_neighbors(None, None, None, -1, None, 0, 0, 0, None, None, None) == 1
    g = mod.GatArgs()
    g.H, g.d, g.dv, g.n_dst, g.scale, g.drop_rate = 2, 4, 4, 3, 2.0, 1.5
    assert lib.tfgx_gat_fused_f32(ctypes.byref(g), None) == 1
    kept = sum(lib.tfgx_dropout_keep(0x1234567890, i, 0.25) for i in range(2000))
    assert 1400 < kept < 1600
    checked += 19
    print("ASAN_HOST_CHECK_OK {} checks on {}".format(checked, path))


if __name__ == "__main__":
    main(sys.argv[1])
