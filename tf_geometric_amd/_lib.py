# coding=utf-8
"""ctypes binding of the C ABI declared in include/tfgx.h (libtfgx.so, hand-written HIP for gfx950).

There is NO CPU fallback: if the library is missing, or there is no GPU, every operator raises.
PyTorch is used only as plumbing: device memory (torch.Tensor), streams, torch.distributed.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFGX_LIB_PATH") or os.path.join(_HERE, "lib", "libtfgx.so")   # (override: developer A/B of kernel builds)

SUM, MEAN, MAX = 0, 1, 2
ACT_NONE, ACT_RELU = 0, 1
NORM_BOTH, NORM_LEFT, NORM_RIGHT = 0, 1, 2
NORM_MODES = {"both": NORM_BOTH, "left": NORM_LEFT, "right": NORM_RIGHT}

c_i32p = ctypes.c_void_p
c_f32p = ctypes.c_void_p


class TfgxError(RuntimeError):
    pass


ABI_VERSION = 114      # include/tfgx.h TFGX_ABI_VERSION: struct layouts / signatures bound below


class ReduceArgs(ctypes.Structure):
    """struct tfgx_reduce_args (include/tfgx.h)."""
    _fields_ = [
        ("row_begin", ctypes.c_void_p), ("row_end", ctypes.c_void_p), ("rp_stride", ctypes.c_int64),
        ("col", ctypes.c_void_p), ("w", ctypes.c_void_p), ("n_dst", ctypes.c_int64),
        ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("F", ctypes.c_int64),
        ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64),
        ("op", ctypes.c_int32), ("act", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("hub_threshold", ctypes.c_int32),
        ("self_coef", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("add_x", ctypes.c_void_p),
        ("ld_add", ctypes.c_int64), ("mean_count", ctypes.c_void_p),
        ("hub_rows", ctypes.c_void_p), ("hub_chunk_ptr", ctypes.c_void_p), ("hub_chunk_begin", ctypes.c_void_p),
        ("hub_chunk_end", ctypes.c_void_p), ("n_hub_rows", ctypes.c_int64), ("n_hub_chunks", ctypes.c_int64),
        ("hub_scratch", ctypes.c_void_p),
        ("x_tail", ctypes.c_void_p), ("ld_tail", ctypes.c_int64), ("f_main", ctypes.c_int64),
        ("edge_tail", ctypes.c_void_p), ("ld_edge_tail", ctypes.c_int64),
        ("row_order", ctypes.c_void_p),
        ("track", ctypes.c_void_p), ("ld_track", ctypes.c_int64), ("track_row_begin", ctypes.c_void_p),
        ("hub_order_slot", ctypes.c_void_p),
        ("wide_blocks", ctypes.c_int32), ("reserved_r5", ctypes.c_int32),
    ]


class GatArgs(ctypes.Structure):
    """struct tfgx_gat_args (include/tfgx.h)."""
    _fields_ = [
        ("row_ptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("n_dst", ctypes.c_int64),
        ("q", ctypes.c_void_p), ("ldq", ctypes.c_int64),
        ("k", ctypes.c_void_p), ("ldk", ctypes.c_int64),
        ("v", ctypes.c_void_p), ("ldv", ctypes.c_int64),
        ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64),
        ("H", ctypes.c_int32), ("d", ctypes.c_int32), ("dv", ctypes.c_int32), ("add_self_loop", ctypes.c_int32),
        ("scale", ctypes.c_float), ("act", ctypes.c_int32), ("bias", ctypes.c_void_p),
        ("row_begin", ctypes.c_void_p), ("row_end", ctypes.c_void_p), ("rp_stride", ctypes.c_int64),
        ("state_acc", ctypes.c_void_p), ("state_ml", ctypes.c_void_p),
        ("hub_threshold", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("hub_rows", ctypes.c_void_p), ("hub_chunk_ptr", ctypes.c_void_p), ("hub_chunk_begin", ctypes.c_void_p),
        ("hub_chunk_end", ctypes.c_void_p), ("hub_chunk_row", ctypes.c_void_p),
        ("n_hub_rows", ctypes.c_int64), ("n_hub_chunks", ctypes.c_int64),
        ("hub_scratch_acc", ctypes.c_void_p), ("hub_scratch_ml", ctypes.c_void_p),
        ("stats_ml", ctypes.c_void_p),
        ("drop_rate", ctypes.c_float), ("reserved2", ctypes.c_int32), ("drop_seed", ctypes.c_uint64),
        ("drop_self_base", ctypes.c_int64),
        ("row_order", ctypes.c_void_p),
        ("drop_seed_dev", ctypes.c_void_p),
        ("state_in_acc", ctypes.c_void_p), ("state_in_ml", ctypes.c_void_p),
        ("qgrad_t", ctypes.c_void_p), ("qgrad_s", ctypes.c_void_p), ("state_t", ctypes.c_void_p), ("state_s", ctypes.c_void_p),
        ("state_in_t", ctypes.c_void_p), ("state_in_s", ctypes.c_void_p),
    ]


class HubLists(ctypes.Structure):
    """struct tfgx_hub_lists (include/tfgx.h)."""
    _fields_ = [("threshold", ctypes.c_int32), ("reserved", ctypes.c_int32), ("n_rows", ctypes.c_int64),
                ("n_chunks", ctypes.c_int64), ("rows", ctypes.c_void_p), ("chunk_ptr", ctypes.c_void_p),
                ("chunk_begin", ctypes.c_void_p), ("chunk_end", ctypes.c_void_p), ("chunk_row", ctypes.c_void_p)]


def edge_softmax(plan, score, H, out):
    """tfgx_edge_softmax_hub_f32 on `plan` (hub rows chunk-wise when the plan has any)."""
    lib = require_gpu()
    hub, nc = hub_lists(plan)
    hp = 1
    while hp < H:
        hp <<= 1
    scratch = torch.empty(max(nc * 2 * hp, 1), dtype=torch.float32, device=out.device) if hub is not None else None
    check(lib.tfgx_edge_softmax_hub_f32(ptr(plan.row_ptr), ptr(plan.perm), ptr(score), H, plan.n_dst, ptr(out),
                                        None if hub is None else ctypes.byref(hub), ptr(scratch), stream_ptr()),
          "tfgx_edge_softmax_hub_f32")
    return out


def hub_lists(plan):
    """(HubLists struct, n_chunks) of a plan's long rows, or (None, 0) when the plan has none (plan.hub_info())."""
    info = plan.hub_info()
    if info is None:
        return None, 0
    hub_rows, chunk_ptr, chunk_begin, chunk_end, chunk_row = info
    h = HubLists()
    h.threshold, h.n_rows, h.n_chunks = int(plan.hub_threshold), int(hub_rows.shape[0]), int(chunk_begin.shape[0])
    h.rows, h.chunk_ptr = hub_rows.data_ptr(), chunk_ptr.data_ptr()
    h.chunk_begin, h.chunk_end, h.chunk_row = chunk_begin.data_ptr(), chunk_end.data_ptr(), chunk_row.data_ptr()
    return h, int(chunk_begin.shape[0])


class GatBackwardArgs(ctypes.Structure):
    """struct tfgx_gat_backward_args (include/tfgx.h)."""
    _fields_ = [
        ("row_ptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("n_dst", ctypes.c_int64),
        ("row_ptr_t", ctypes.c_void_p), ("dst_t", ctypes.c_void_p), ("n_src", ctypes.c_int64),
        ("q", ctypes.c_void_p), ("ldq", ctypes.c_int64),
        ("k", ctypes.c_void_p), ("ldk", ctypes.c_int64),
        ("v", ctypes.c_void_p), ("ldv", ctypes.c_int64),
        ("grad_out", ctypes.c_void_p), ("ld_grad_out", ctypes.c_int64),
        ("stats_ml", ctypes.c_void_p), ("dsum", ctypes.c_void_p),
        ("H", ctypes.c_int32), ("d", ctypes.c_int32), ("dv", ctypes.c_int32), ("add_self_loop", ctypes.c_int32),
        ("scale", ctypes.c_float), ("reserved", ctypes.c_int32),
        ("grad_q", ctypes.c_void_p), ("ld_grad_q", ctypes.c_int64),
        ("grad_k", ctypes.c_void_p), ("ld_grad_k", ctypes.c_int64),
        ("grad_v", ctypes.c_void_p), ("ld_grad_v", ctypes.c_int64),
        ("drop_rate", ctypes.c_float), ("reserved2", ctypes.c_int32), ("drop_seed", ctypes.c_uint64),
        ("drop_self_base", ctypes.c_int64), ("edge_pos_t", ctypes.c_void_p),
        ("ld_stats_ml", ctypes.c_int64), ("ld_dsum", ctypes.c_int64),
        ("row_order", ctypes.c_void_p), ("row_order_t", ctypes.c_void_p),
        ("drop_seed_dev", ctypes.c_void_p),
        ("span_begin", ctypes.c_void_p), ("span_end", ctypes.c_void_p), ("span_stride", ctypes.c_int64),
        ("accumulate", ctypes.c_int32), ("reserved3", ctypes.c_int32),
        ("head_pack", ctypes.c_void_p), ("ld_head_pack", ctypes.c_int64),
    ]


# name -> (restype, argtypes); this table is checked against include/tfgx.h by tests/test_abi.py
_I64, _I32, _F32, _P, _SZ = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
SIGNATURES = {
    "tfgx_version": (ctypes.c_int, []),
    "tfgx_column_sum_workspace_bytes": (_SZ, [_I64, _I64]),
    "tfgx_column_sum_f32": (ctypes.c_int, [_P, _I64, _I64, _I64, _P, _P, _SZ, _P]),
    "tfgx_last_error": (ctypes.c_char_p, []),
    "tfgx_csr_plan_workspace_bytes": (_SZ, [_I64, _I64]),
    "tfgx_build_csr_by_dst": (ctypes.c_int, [_P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    "tfgx_merge_edges_workspace_bytes": (_SZ, [_I64, _I64]),
    "tfgx_merge_duplicated_edges": (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _SZ, _P]),
    "tfgx_segment_topk_workspace_bytes": (_SZ, [_I64, _I64]),
    "tfgx_segment_topk": (ctypes.c_int, [_P, _P, _I64, _I64, _I32, _F32, _P, _P, _P, _SZ, _P]),
    "tfgx_segment_max_with_count_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P]),
    "tfgx_segment_max_with_arg_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _P]),
    "tfgx_segment_max_backward_mask_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "tfgx_pool_mlp_max_wgrad_applies": (ctypes.c_int, [_I64, _I64]),
    "tfgx_pool_mlp_max_wgrad_plan_bytes": (_SZ, [_I64, _I64, _I64, _I64]),
    "tfgx_pool_mlp_max_wgrad_plan": (ctypes.c_int, [_P, _I64, _I64, _I64, _I64, _P, _SZ, _P]),
    "tfgx_pool_mlp_max_wgrad_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "tfgx_pool_mlp_max_wgrad_f32": (ctypes.c_int, [_P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64,
                                                   _P, _P, _I64, _P, _P, _SZ, _P]),
    "tfgx_segment_max_backward_mask_f32": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P,
                                                          _I64, _P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _P, _SZ, _P]),
    "tfgx_segment_max_backward_mask_phases_f32": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P,
                                                                 _I64, _P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _P, _SZ,
                                                                 ctypes.c_int32, _P]),
    "tfgx_segment_max_backward_push_f32": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P,
                                                          _I64, _P, _I64, _P, _I64, _P]),
    "tfgx_segment_max_backward_w_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _P]),
    "tfgx_gemm_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "tfgx_gemm_tn_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I32]),
    "tfgx_gemm_tn_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _P, _P, _SZ, _P]),
    "tfgx_gemm_tn_gated_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _P, _P, _SZ, _P]),
    "tfgx_transpose_f32": (ctypes.c_int, [_P, _I64, _I64, _I64, _P, _I64, _P]),
    "tfgx_gemm_bias_act_cols_ws_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I32, _I64, _P, _I64, _I64, _I64, _I64, _P, _SZ, _P]),
    "tfgx_dropout_keep": (ctypes.c_int32, [ctypes.c_uint64, ctypes.c_uint32, _F32]),
    "tfgx_permute_rows_f32": (ctypes.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "tfgx_gat_pack_dst_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _P, _P]),
    "tfgx_gat_query_grad_d1_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, ctypes.c_float, _P, _I64, _P]),
    "tfgx_gat_pack_dst_heads_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _P, _P]),
    "tfgx_relu_backward_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _I64, _P, _I64, _P]),
    "tfgx_scatter_add_rows_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "tfgx_segment_reduce_f32": (ctypes.c_int, [ctypes.POINTER(ReduceArgs), _P]),
    "tfgx_segment_reduce_describe": (ctypes.c_int, [ctypes.POINTER(ReduceArgs), ctypes.c_char_p, ctypes.c_size_t]),
    "tfgx_segment_weight_sum_f32": (ctypes.c_int, [_P, _P, _I64, _F32, _P, _P]),
    "tfgx_gcn_norm_edges_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _P, _I32, _F32, _I32, _I32, _P, _P, _P]),
    "tfgx_edge_softmax_f32": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _P]),
    "tfgx_gat_fused_f32": (ctypes.c_int, [ctypes.POINTER(GatArgs), _P]),
    "tfgx_gat_merge_passes_f32": (ctypes.c_int, [ctypes.POINTER(GatArgs), _P, _P, _I32, _P]),
    "tfgx_gat_merge_parts_f32": (ctypes.c_int, [ctypes.POINTER(GatArgs), _P, _P, _P, _P, _P]),
    "tfgx_sddmm_f32": (ctypes.c_int, [_P, _P, _I64, _P, _I64, _P, _I64, _I64, _P, _P]),
    "tfgx_segment_max_count_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P]),
    "tfgx_segment_max_backward_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64,
                                                     _P, _I64, _I64, _P, _P]),
    "tfgx_gat_backward_dst_f32": (ctypes.c_int, [ctypes.POINTER(GatBackwardArgs), _P]),
    "tfgx_gat_backward_src_f32": (ctypes.c_int, [ctypes.POINTER(GatBackwardArgs), _P]),
    "tfgx_edge_softmax_hub_f32": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, ctypes.POINTER(HubLists), _P, _P]),
    "tfgx_sddmm_hub_f32": (ctypes.c_int, [_P, _P, _I64, _P, _I64, _P, _I64, _I64, _P, ctypes.POINTER(HubLists), _P]),
    "tfgx_gat_backward_dst_hub_f32": (ctypes.c_int, [ctypes.POINTER(GatBackwardArgs), ctypes.POINTER(HubLists), _P, _P]),
    "tfgx_gat_backward_src_hub_f32": (ctypes.c_int, [ctypes.POINTER(GatBackwardArgs), ctypes.POINTER(HubLists), _P, _P]),
    "tfgx_segment_max_count_hub_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64,
                                                      ctypes.POINTER(HubLists), _P, _P]),
    "tfgx_segment_max_backward_hub_f32": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _I64, _P, _I64,
                                                         _P, _I64, _I64, _P, ctypes.POINTER(HubLists), _P, _P]),
    "tfgx_head_mean_f32": (ctypes.c_int, [_P, _I64, _I64, _I32, _I32, _P, _I32, _P, _I64, _P]),
    "tfgx_gemm_bias_act_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I32, _P, _I64, _I64, _I64, _I64, _P]),
    "tfgx_gemm_bias_act_cols_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _I32, _I64, _P, _I64, _I64, _I64, _I64, _P]),
    "tfgx_l2_normalize_rows_f32": (ctypes.c_int, [_P, _I64, _I64, _I64, _P]),
    "tfgx_sample_neighbors": (ctypes.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, ctypes.c_uint64, _P, _P, _P]),
    "tfgx_split_rows_f32": (ctypes.c_int, [_P, _I64, _I64, _I64, _I64, _P, _I64, _P, _I64, _P]),
    "tfgx_split_rows_verify_f32": (ctypes.c_int, [_P, _I64, _I64, _I64, _I64, _P, _I64, _P, _I64, _I64, ctypes.c_uint64, _P, _P]),
    "tfgx_gather_rows_f32": (ctypes.c_int, [_P, _I64, _P, _I64, _I64, _P, _I64, _P]),
    "tfgx_halo_workspace_bytes": (_SZ, [_I64]),
    "tfgx_halo_mark": (ctypes.c_int, [_P, _I64, _I32, _I32, _I64, _P, _P]),
    "tfgx_halo_compact": (ctypes.c_int, [_P, _I64, _P, _P, _P, _P, _SZ, _P]),
    "tfgx_halo_remap_cols": (ctypes.c_int, [_P, _I64, _I32, _I32, _P, _I32, _P, _P]),
    "tfgx_split_by_source_class": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _P, _I32, _P, _P, _P, _P]),
    "tfgx_aggregate_gemm_fits": (ctypes.c_int, [_I64, _I64]),
    "tfgx_aggregate_gemm_f32": (ctypes.c_int, [ctypes.POINTER(ReduceArgs), _P, _I64, _P, _I32, _P, _I64, _I64, _P]),
}

_lib = None


def load_library():
    """Load libtfgx.so and bind every entry point of include/tfgx.h. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        try:     # sources are in-tree: build once (hipcc --offload-arch=gfx950, ~15 s); never a CPU fallback
            from . import _build
            _build.build(verbose=False)
        except Exception as ex:      # noqa: BLE001 - reported below with the build hint
            raise TfgxError("tf_geometric_amd: building {} failed: {}".format(LIB_PATH, ex))
    if not os.path.exists(LIB_PATH):
        raise TfgxError(
            "tf_geometric_amd: HIP library {} is missing. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.tfgx_version() != ABI_VERSION:
        raise TfgxError("tf_geometric_amd: {} was built for ABI {} but this package binds ABI {} (include/tfgx.h "
                        "TFGX_ABI_VERSION): rebuild with __graft_entry__.build()".format(LIB_PATH, lib.tfgx_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load_library().tfgx_last_error()
        raise TfgxError("{} failed with code {}: {}".format(what or "tfgx call", rc, (msg or b"").decode()))


def require_gpu():
    if not torch.cuda.is_available():
        raise TfgxError("tf_geometric_amd needs an AMD GPU (gfx950 / MI355X); torch.cuda.is_available() is False "
                        "and there is no CPU fallback.")
    return load_library()


def device():
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def as_f32(x, dev=None):
    """numpy / list / torch -> contiguous float32 tensor on the GPU (float64 is down-cast as data/graph.py:79-86)."""
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        # keep the autograd edge of tensors that are being trained (autograd.py); plain inputs are detached
        t = x if (x.requires_grad and torch.is_grad_enabled()) else x.detach()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if t.device != dev:
        t = t.to(dev)
    if t.dim() >= 2 and t.stride(-1) != 1:
        t = t.contiguous()
    if t.dim() == 1 and not t.is_contiguous():
        t = t.contiguous()
    return t


def as_i32(x, dev=None):
    dev = dev or device()
    if isinstance(x, torch.Tensor):
        t = x.detach()
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype != torch.int32:
        t = t.to(torch.int32)          # data/graph.py:59-66: edge_index is cast to int32
    if t.device != dev:
        t = t.to(dev)
    return t.contiguous()


def row_major_2d(t):
    """Return (tensor, leading dimension) for a 2-D float32 tensor whose rows are dense."""
    assert t.dim() == 2
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, ld
