# coding=utf-8
"""Python side of the tf.load_op_library route (BASELINE.json north_star) — NEVER IMPORTED OR RUN HERE: TensorFlow is absent
from this image.  tests/test_abi.py compiles this file (py_compile) and checks that every op it names is registered by
tfgx_tf_ops.cc; the ops themselves ARE executed — against the mock runtime of integration/tf_shim/mock/, in the pipelines
this file composes (tests/test_gpu_tf_shim.py) — but nothing here can run without TensorFlow.

What it holds, for a maintainer with a TensorFlow-ROCm build:
  * `load()`                      tf.load_op_library(libtfgx_tf_ops.so) (built by build_tf_shim.sh)
  * @tf.RegisterGradient(...)     the gradients of the forward ops in terms of the backward ops of the same library —
                                  what tf.GradientTape then derives for a layer (demo/demo_gcn.py:68-77)
  * `gcn_layer(...)`              tfg.layers.GCN.call's body ([x, edge_index, edge_weight], layers/conv/gcn.py:129-156) on
                                  the ops: BuildCsrByDst -> GcnNormEdges -> GemmBiasAct -> SegmentReduce
  * `dist_comm(...)`              in-process ncclComm_t through ctypes on libtfgx_dist.so (tfgx_dist_unique_id /
                                  tfgx_dist_comm_init), the handle the TfgxHalo* ops take as an int64 scalar
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_ops = None


def load(path=None):
    """tf.load_op_library of the shim; the returned module exposes tfgx_segment_reduce, tfgx_gemm_bias_act, ..."""
    global _ops
    if _ops is None:
        import tensorflow as tf
        _ops = tf.load_op_library(path or os.path.join(_HERE, "libtfgx_tf_ops.so"))
        _register_gradients(tf, _ops)
    return _ops


def _transposed_plan(tf, ops, row_ptr, col, perm, num_nodes):
    """CSR-by-source plan of the same edges (the plan d/dx runs on), from the forward plan: the row of every CSR position
    is recovered from row_ptr, (col, row) swapped and bucketed again.  -> (row_ptr_t, dst_t, perm_t)."""
    deg = row_ptr[1:] - row_ptr[:-1]
    rows = tf.repeat(tf.range(num_nodes, dtype=tf.int32), deg)
    return ops.tfgx_build_csr_by_dst(edge_index=tf.stack([col, rows]), num_nodes=num_nodes)


def _register_gradients(tf, ops):
    @tf.RegisterGradient("TfgxSegmentReduce")
    def _segment_reduce_grad(op, g):
        row_ptr, col, w, x, self_coef, bias = op.inputs
        kind, act = op.get_attr("op"), op.get_attr("act")
        if kind == 2:
            raise NotImplementedError("max: use the tfgx_segment_max_* entry points (include/tfgx.h:263-316)")
        if act == 1:
            g = ops.tfgx_relu_backward(g=g, out=op.outputs[0])
        n = tf.shape(x)[0]
        if kind == 1:     # mean: the divisor is the in-degree (max(count, 1))
            deg = tf.cast(tf.maximum(row_ptr[1:] - row_ptr[:-1], 1), g.dtype)
            g = g / deg[:, None]
        row_ptr_t, dst_t, perm_t = _transposed_plan(tf, ops, row_ptr, col, None, n)
        empty = tf.zeros([0], g.dtype)
        has_w = tf.size(w) > 0
        w_t = tf.cond(has_w, lambda: ops.tfgx_permute_rows(src=w, perm=perm_t), lambda: empty)
        # d/dx = the forward kernel on the transposed plan (+ the implicit self-loop term)
        gx = ops.tfgx_segment_reduce(row_ptr=row_ptr_t, col=dst_t, w=w_t, x=g, self_coef=self_coef, bias=empty, op=0, act=0)
        gw = tf.cond(has_w, lambda: ops.tfgx_sddmm(row_ptr=row_ptr, col=col, a=g, b=x), lambda: empty)
        g_sc = tf.cond(tf.size(self_coef) > 0, lambda: tf.reduce_sum(g * x, axis=1), lambda: empty)
        g_bias = tf.cond(tf.size(bias) > 0, lambda: tf.reduce_sum(g, axis=0), lambda: empty)
        return [None, None, gw, gx, g_sc, g_bias]

    @tf.RegisterGradient("TfgxGemmBiasAct")
    def _gemm_grad(op, g):
        x, kernel, bias = op.inputs
        if op.get_attr("act") == 1:
            g = ops.tfgx_relu_backward(g=g, out=op.outputs[0])
        dw, db = ops.tfgx_gemm_tn(x=x, g=g)                                   # x^T @ g and the column sums of g
        dx = ops.tfgx_gemm_bias_act(x=g, kernel=tf.transpose(kernel), bias=tf.zeros([0], g.dtype), act=0)
        return [dx, dw, tf.cond(tf.size(bias) > 0, lambda: db, lambda: tf.zeros([0], g.dtype))]

    @tf.RegisterGradient("TfgxAggregateGemm")
    def _aggregate_gemm_grad(op, g, g_agg_unused):
        # out = act(aggregate @ kernel + bias), aggregate = reduce(x) returned as the op's second output (want_aggregate=True)
        row_ptr, col, w, x, self_coef, kernel, bias = op.inputs
        if not op.get_attr("want_aggregate"):
            raise NotImplementedError("TfgxAggregateGemm under a GradientTape needs want_aggregate=True")
        if op.get_attr("act") == 1:
            g = ops.tfgx_relu_backward(g=g, out=op.outputs[0])
        empty = tf.zeros([0], g.dtype)
        dw, db = ops.tfgx_gemm_tn(x=op.outputs[1], g=g)                         # aggregate^T @ g, column sums of g
        d_agg = ops.tfgx_gemm_bias_act(x=g, kernel=tf.transpose(kernel), bias=empty, act=0)
        n = tf.shape(x)[0]
        if op.get_attr("op") == 1:
            d_agg = d_agg / tf.cast(tf.maximum(row_ptr[1:] - row_ptr[:-1], 1), g.dtype)[:, None]
        row_ptr_t, dst_t, perm_t = _transposed_plan(tf, ops, row_ptr, col, None, n)
        has_w = tf.size(w) > 0
        w_t = tf.cond(has_w, lambda: ops.tfgx_permute_rows(src=w, perm=perm_t), lambda: empty)
        gx = ops.tfgx_segment_reduce(row_ptr=row_ptr_t, col=dst_t, w=w_t, x=d_agg, self_coef=self_coef, bias=empty, op=0, act=0)
        gw = tf.cond(has_w, lambda: ops.tfgx_sddmm(row_ptr=row_ptr, col=col, a=d_agg, b=x), lambda: empty)
        g_sc = tf.cond(tf.size(self_coef) > 0, lambda: tf.reduce_sum(d_agg * x, axis=1), lambda: empty)
        return [None, None, gw, gx, g_sc, dw, tf.cond(tf.size(bias) > 0, lambda: db, lambda: empty)]

    @tf.RegisterGradient("TfgxHaloExchange")
    def _halo_grad(op, g_table):
        x_own, send_idx, sc, rc, ds, comm = op.inputs
        d_own = ops.tfgx_halo_reverse(d_table=g_table, send_idx=send_idx, send_counts=sc, recv_counts=rc,
                                      send_dense_start=ds, comm=comm, world=op.get_attr("world"),
                                      rank=op.get_attr("rank"), rounds=op.get_attr("rounds"), n_own=tf.shape(x_own)[0])
        return [d_own, None, None, None, None, None]

    @tf.RegisterGradient("TfgxSddmm")
    def _sddmm_grad(op, g):
        raise NotImplementedError("second-order gradients are not provided")


def gcn_layer(x, edge_index, edge_weight, kernel, bias, num_nodes, activation_is_relu=False, cache=None,
              inference_only=False):
    """tfg.layers.GCN.call on the ops (norm='both', renorm=True: the layer's defaults, layers/conv/gcn.py:32-40).
    `cache`: the dict the reference keeps its normalised adjacency in (nn/conv/gcn.py:125-128); here it keeps the plan."""
    import tensorflow as tf
    ops = load()
    plan = None if cache is None else cache.get("tfgx_plan")
    if plan is None:
        row_ptr, col, perm = ops.tfgx_build_csr_by_dst(edge_index=edge_index, num_nodes=num_nodes)
        w_csr = tf.zeros([0], tf.float32) if edge_weight is None else ops.tfgx_permute_rows(src=edge_weight, perm=perm)
        w_norm, self_coef = ops.tfgx_gcn_norm_edges(row_ptr=row_ptr, col=col, w=w_csr)
        plan = (row_ptr, col, w_norm, self_coef)
        if cache is not None:
            cache["tfgx_plan"] = plan
    row_ptr, col, w_norm, self_coef = plan
    empty = tf.zeros([0], tf.float32)
    f_in, units = int(x.shape[-1]), int(kernel.shape[-1])
    fits = units > f_in and f_in % 4 == 0 and 4 <= f_in <= 128 and units <= 256      # tfgx_aggregate_gemm_fits (ABI 110)
    if fits:
        # (A_hat x) W in ONE launch; under a GradientTape the same launch also returns the aggregate (d/dkernel needs it)
        out, _ = ops.tfgx_aggregate_gemm(row_ptr=row_ptr, col=col, w=w_norm, x=x, self_coef=self_coef, kernel=kernel,
                                         bias=empty if bias is None else bias, op=0, act=1 if activation_is_relu else 0,
                                         want_aggregate=not inference_only)
        return out
    h = ops.tfgx_gemm_bias_act(x=x, kernel=kernel, bias=empty, act=0)            # gcn.py:272
    return ops.tfgx_segment_reduce(row_ptr=row_ptr, col=col, w=w_norm, x=h, self_coef=self_coef,
                                   bias=empty if bias is None else bias, op=0, act=1 if activation_is_relu else 0)


def dist_comm(world, rank, broadcast_bytes):
    """ncclComm_t for the TfgxHalo* / TfgxAllReduceSum ops: rank 0 draws the unique id, `broadcast_bytes(b)` is the host's
    control channel (returns rank 0's 128 bytes on every rank), every rank initialises.  -> int (the pointer value)."""
    lib = ctypes.CDLL(os.path.join(_ROOT, "tf_geometric_amd", "lib", "libtfgx_dist.so"))
    uid = ctypes.create_string_buffer(128)
    if rank == 0 and lib.tfgx_dist_unique_id(uid) != 0:
        raise RuntimeError("tfgx_dist_unique_id failed")
    uid = ctypes.create_string_buffer(broadcast_bytes(bytes(uid.raw)), 128)
    comm = ctypes.c_void_p()
    lib.tfgx_dist_comm_init.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    if lib.tfgx_dist_comm_init(world, rank, uid, ctypes.byref(comm)) != 0:
        raise RuntimeError("tfgx_dist_comm_init failed")
    return int(comm.value)
