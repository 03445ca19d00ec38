// TensorFlow custom-op shim over the tfgx C ABI (include/tfgx.h) — the `tf.load_op_library` route BASELINE.json's
// north_star names.  A SKETCH, NEVER LINKED OR RUN: TensorFlow (and tf_sparse) are absent from the image and cannot be
// installed (no network).  What IS checked here: the file type-checks (`g++ -fsyntax-only`) against include/tfgx.h and
// a mock of the four TensorFlow headers it uses (integration/tf_shim/mock/, tests/test_abi.py) — so every C-ABI call
// below has the argument types the header declares.  A maintainer with a TensorFlow-ROCm build compiles it with
// build_tf_shim.sh; the executable, tested boundary of this repo is the ctypes host in tf_geometric_amd/ and the
// torch-free C++ programs in examples/.  Each op forwards raw device pointers + sizes to C-ABI calls on TF's own HIP
// stream.  Gradients are registered on the Python side (integration/tf_shim/tfgx_tf.py) in terms of the same ops.
//
//   TfgxBuildCsrByDst   edge_index[2,E] -> row_ptr[N+1], col[E], perm[E]          (tfgx_build_csr_by_dst)
//   TfgxSegmentReduce   plan + x[N,F] (+ w[E] in CSR order) -> out[N,F]           (tfgx_segment_reduce_f32)
//                       replaces tf.gather + gcn_mapper + tf.math.unsorted_segment_{sum,mean,max}
//                       (tf_geometric/nn/kernel/map_reduce.py:15-42,60-70)
//   TfgxGatFused        plan + Q,K,V -> out                                       (tfgx_gat_fused_f32)
//                       replaces tf_geometric/nn/conv/gat.py:56-89,112
//   TfgxGcnNormEdges    plan + raw weights -> normalised weights (CSR order) + self_coef   (tfgx_segment_weight_sum_f32 +
//                       tfgx_gcn_norm_edges_f32; gcn_norm_adj, nn/conv/gcn.py:32-130, sym=True)
//   TfgxGemmBiasAct     act(x @ kernel + bias) on the fp32 matrix cores          (tfgx_gemm_bias_act_f32; gcn.py:272,284-288)
// With these five a tfg.layers.GCN call ([x, edge_index, edge_weight], layers/conv/gcn.py:129-156) is:
//   BuildCsrByDst (cached) -> GcnNormEdges (cached) -> GemmBiasAct(x, kernel) -> SegmentReduce(sum, w, self_coef, bias, act)
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"
#include <cmath>
#include "tfgx.h"

using namespace tensorflow;

namespace {
inline tfgx_stream_t TfStream(OpKernelContext* ctx) {
  return reinterpret_cast<tfgx_stream_t>(ctx->eigen_gpu_device().stream());   // hipStream_t on TF-ROCm
}
}  // namespace

REGISTER_OP("TfgxBuildCsrByDst")
    .Input("edge_index: int32")
    .Attr("num_nodes: int")
    .Output("row_ptr: int32")
    .Output("col: int32")
    .Output("perm: int32");

class TfgxBuildCsrByDstOp : public OpKernel {
 public:
  explicit TfgxBuildCsrByDstOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("num_nodes", &n_)); }
  void Compute(OpKernelContext* ctx) override {
    const Tensor& ei = ctx->input(0);
    OP_REQUIRES(ctx, ei.dims() == 2 && ei.dim_size(0) == 2, errors::InvalidArgument("edge_index must be [2, E]"));
    const int64_t E = ei.dim_size(1);
    Tensor *row_ptr, *col, *perm, ws;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n_ + 1}, &row_ptr));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, {E}, &col));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(2, {E}, &perm));
    const size_t ws_bytes = tfgx_csr_plan_workspace_bytes(n_, E);
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_UINT8, {static_cast<int64_t>(ws_bytes)}, &ws));
    const int32_t* e = ei.flat<int32>().data();
    const int rc = tfgx_build_csr_by_dst(e, e + E, E, n_, n_, row_ptr->flat<int32>().data(), col->flat<int32>().data(),
                                         perm->flat<int32>().data(), ws.flat<uint8>().data(), ws_bytes, TfStream(ctx));
    OP_REQUIRES(ctx, rc == 0, errors::InvalidArgument(tfgx_last_error()));   // out-of-range id, like TF-CPU
  }
  int64_t n_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxBuildCsrByDst").Device(DEVICE_GPU), TfgxBuildCsrByDstOp);

REGISTER_OP("TfgxSegmentReduce")
    .Input("row_ptr: int32")
    .Input("col: int32")
    .Input("w: float")          // [E] in CSR order, or [0] for the unweighted (identity_mapper) path
    .Input("x: float")
    .Input("self_coef: float")  // [N] weight of the implicit (r, r) edge (GCN's added diagonal), or [0]
    .Input("bias: float")       // [F] or [0]
    .Attr("op: int")            // 0 sum, 1 mean, 2 max
    .Attr("act: int = 0")       // 0 none, 1 relu
    .Output("out: float");

class TfgxSegmentReduceOp : public OpKernel {
 public:
  explicit TfgxSegmentReduceOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("op", &op_));
    OP_REQUIRES_OK(c, c->GetAttr("act", &act_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &w = ctx->input(2), &x = ctx->input(3);
    const Tensor &sc = ctx->input(4), &bias = ctx->input(5);
    const int64_t n = rp.dim_size(0) - 1, F = x.dim_size(1);
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n, F}, &out));
    tfgx_reduce_args a = {};
    a.row_begin = rp.flat<int32>().data();
    a.row_end = a.row_begin + 1;
    a.rp_stride = 1;
    a.col = col.flat<int32>().data();
    a.w = w.NumElements() ? w.flat<float>().data() : nullptr;
    a.n_dst = n;
    a.x = x.flat<float>().data();
    a.ldx = F;
    a.F = F;
    a.out = out->flat<float>().data();
    a.ldo = F;
    a.op = op_;
    a.act = act_;
    a.self_coef = sc.NumElements() ? sc.flat<float>().data() : nullptr;
    a.bias = bias.NumElements() ? bias.flat<float>().data() : nullptr;
    OP_REQUIRES(ctx, tfgx_segment_reduce_f32(&a, TfStream(ctx)) == 0, errors::Internal(tfgx_last_error()));
  }
  int op_, act_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxSegmentReduce").Device(DEVICE_GPU), TfgxSegmentReduceOp);

REGISTER_OP("TfgxGatFused")
    .Input("row_ptr: int32")
    .Input("col: int32")
    .Input("q: float")
    .Input("k: float")
    .Input("v: float")
    .Attr("num_heads: int")
    .Output("out: float");

class TfgxGatFusedOp : public OpKernel {
 public:
  explicit TfgxGatFusedOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("num_heads", &h_)); }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &q = ctx->input(2), &k = ctx->input(3), &v = ctx->input(4);
    const int64_t n = rp.dim_size(0) - 1, A = q.dim_size(1), W = v.dim_size(1);
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n, W}, &out));
    tfgx_gat_args a = {};
    a.row_ptr = rp.flat<int32>().data();
    a.col = col.flat<int32>().data();
    a.n_dst = n;
    a.q = q.flat<float>().data(); a.ldq = A;
    a.k = k.flat<float>().data(); a.ldk = A;
    a.v = v.flat<float>().data(); a.ldv = W;
    a.out = out->flat<float>().data(); a.ldo = W;
    a.H = h_; a.d = static_cast<int32_t>(A / h_); a.dv = static_cast<int32_t>(W / h_);
    a.add_self_loop = 1;                                  // gat.py:43
    a.scale = std::sqrt(static_cast<float>(A / h_));      // gat.py:78
    OP_REQUIRES(ctx, tfgx_gat_fused_f32(&a, TfStream(ctx)) == 0, errors::Internal(tfgx_last_error()));
  }
  int h_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxGatFused").Device(DEVICE_GPU), TfgxGatFusedOp);

REGISTER_OP("TfgxGcnNormEdges")
    .Input("row_ptr: int32")
    .Input("col: int32")
    .Input("w: float")               // [E] raw weights in CSR order, or [0] = ones (Graph default, data/graph.py:53-56)
    .Attr("norm: int = 0")           // 0 both, 1 left, 2 right (nn/conv/gcn.py:74,101,111)
    .Attr("add_self_loop: bool = true")
    .Attr("renorm: bool = true")
    .Attr("improved: bool = false")
    .Output("w_out: float")          // [E] normalised weights, CSR order
    .Output("self_coef: float");     // [N] weight of the implicit diagonal entry

class TfgxGcnNormEdgesOp : public OpKernel {
 public:
  explicit TfgxGcnNormEdgesOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("norm", &norm_));
    OP_REQUIRES_OK(c, c->GetAttr("add_self_loop", &add_self_loop_));
    OP_REQUIRES_OK(c, c->GetAttr("renorm", &renorm_));
    OP_REQUIRES_OK(c, c->GetAttr("improved", &improved_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &w = ctx->input(2);
    const int64_t n = rp.dim_size(0) - 1, E = col.dim_size(0);
    Tensor *w_out = nullptr, *self_coef = nullptr, deg;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {E}, &w_out));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, {n}, &self_coef));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_FLOAT, {n}, &deg));
    const float fill = improved_ ? 2.0f : 1.0f;                                  // gcn.py:62
    const float diag = norm_ == 0 ? ((add_self_loop_ && renorm_) ? fill : 0.0f)  // :76-77
                                  : (add_self_loop_ ? fill : 0.0f);              // :71-72
    const float* wp = w.NumElements() ? w.flat<float>().data() : nullptr;
    OP_REQUIRES(ctx, tfgx_segment_weight_sum_f32(rp.flat<int32>().data(), wp, n, diag, deg.flat<float>().data(),
                                                 TfStream(ctx)) == 0, errors::Internal(tfgx_last_error()));
    OP_REQUIRES(ctx, tfgx_gcn_norm_edges_f32(rp.flat<int32>().data(), col.flat<int32>().data(), wp, n,
                                             deg.flat<float>().data(), nullptr /* sym=True */, norm_, fill,
                                             add_self_loop_ ? 1 : 0, renorm_ ? 1 : 0, w_out->flat<float>().data(),
                                             self_coef->flat<float>().data(), TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
  int norm_;
  bool add_self_loop_, renorm_, improved_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxGcnNormEdges").Device(DEVICE_GPU), TfgxGcnNormEdgesOp);

REGISTER_OP("TfgxGemmBiasAct")
    .Input("x: float")               // [M, K]
    .Input("kernel: float")          // [K, N]
    .Input("bias: float")            // [N] or [0]
    .Attr("act: int = 0")
    .Output("out: float");

class TfgxGemmBiasActOp : public OpKernel {
 public:
  explicit TfgxGemmBiasActOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("act", &act_)); }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &x = ctx->input(0), &k = ctx->input(1), &b = ctx->input(2);
    const int64_t M = x.dim_size(0), K = x.dim_size(1), N = k.dim_size(1);
    OP_REQUIRES(ctx, k.dim_size(0) == K, errors::InvalidArgument("x and kernel do not agree on K"));
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {M, N}, &out));
    // the workspace: split-K partials (small M, long K) or the row kernel's tile counters (tall M); a temp of THIS op call
    Tensor ws;
    const size_t ws_bytes = tfgx_gemm_workspace_bytes(M, K, N);
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_UINT8, {static_cast<int64_t>(ws_bytes)}, &ws));
    OP_REQUIRES(ctx, tfgx_gemm_bias_act_cols_ws_f32(x.flat<float>().data(), K, k.flat<float>().data(), N,
                                                    b.NumElements() ? b.flat<float>().data() : nullptr, act_, N,
                                                    out->flat<float>().data(), N, M, K, N,
                                                    ws_bytes ? static_cast<void*>(ws.flat<uint8_t>().data()) : nullptr, ws_bytes,
                                                    TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
  int act_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxGemmBiasAct").Device(DEVICE_GPU), TfgxGemmBiasActOp);

// The aggregate-then-project layers in ONE launch (tfgx_aggregate_gemm_f32: GCN with units > F evaluated as (A_hat x) W,
// nn/conv/gcn.py:272-288; the neighbour half of mean / sum GraphSAGE, nn/conv/graph_sage.py:34-58).  want_aggregate = true
// (under a GradientTape: d/dkernel = aggregate^T @ g) also returns the [N, F] aggregate — written by the same launch beside
// the projection, which still reads it from LDS; false returns an empty second output.  tfgx_tf.py falls back to
// TfgxSegmentReduce + TfgxGemmBiasAct when tfgx_aggregate_gemm_fits says no.
REGISTER_OP("TfgxAggregateGemm")
    .Input("row_ptr: int32")
    .Input("col: int32")
    .Input("w: float")          // [E] in CSR order, or [0]
    .Input("x: float")          // [N_src, F]
    .Input("self_coef: float")  // [N] or [0]
    .Input("kernel: float")     // [F, U]
    .Input("bias: float")       // [U] or [0]
    .Attr("op: int")            // 0 sum, 1 mean
    .Attr("act: int = 0")
    .Attr("want_aggregate: bool = false")
    .Output("out: float")       // [N, U]
    .Output("aggregate: float");// [N, F] (want_aggregate) or [0]

class TfgxAggregateGemmOp : public OpKernel {
 public:
  explicit TfgxAggregateGemmOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("op", &op_));
    OP_REQUIRES_OK(c, c->GetAttr("act", &act_));
    OP_REQUIRES_OK(c, c->GetAttr("want_aggregate", &want_agg_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &w = ctx->input(2), &x = ctx->input(3);
    const Tensor &sc = ctx->input(4), &k = ctx->input(5), &bias = ctx->input(6);
    const int64_t n = rp.dim_size(0) - 1, F = x.dim_size(1), U = k.dim_size(1);
    OP_REQUIRES(ctx, k.dim_size(0) == F, errors::InvalidArgument("x and kernel do not agree on F"));
    OP_REQUIRES(ctx, tfgx_aggregate_gemm_fits(F, U) == 1,
                errors::InvalidArgument("shape outside tfgx_aggregate_gemm_fits: use TfgxSegmentReduce + TfgxGemmBiasAct"));
    Tensor *out = nullptr, *agg = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n, U}, &out));
    if (want_agg_) OP_REQUIRES_OK(ctx, ctx->allocate_output(1, {n, F}, &agg));
    else OP_REQUIRES_OK(ctx, ctx->allocate_output(1, {0}, &agg));
    tfgx_reduce_args a = {};
    a.row_begin = rp.flat<int32>().data();
    a.row_end = a.row_begin + 1;
    a.rp_stride = 1;
    a.col = col.flat<int32>().data();
    a.w = w.NumElements() ? w.flat<float>().data() : nullptr;
    a.n_dst = n;
    a.x = x.flat<float>().data();
    a.ldx = F;
    a.F = F;
    a.op = op_;
    a.self_coef = sc.NumElements() ? sc.flat<float>().data() : nullptr;
    if (want_agg_) {                                   // side output of the same launch (tfgx.h: args->out)
      a.out = agg->flat<float>().data();
      a.ldo = F;
    }
    OP_REQUIRES(ctx, tfgx_aggregate_gemm_f32(&a, k.flat<float>().data(), U,
                                             bias.NumElements() ? bias.flat<float>().data() : nullptr, act_,
                                             out->flat<float>().data(), U, U, TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
  int op_, act_;
  bool want_agg_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxAggregateGemm").Device(DEVICE_GPU), TfgxAggregateGemmOp);

// ---------------------------------------------------------------------------------------------------------------------
// Backward ops (round 3; still NEVER LINKED OR RUN — see the header of this file).  The gradients are registered on the
// Python side, integration/tf_shim/tfgx_tf.py (@tf.RegisterGradient), in terms of the ops of this file:
//   d/dx of TfgxSegmentReduce (sum / mean)  = TfgxSegmentReduce on the TRANSPOSED plan (TfgxBuildCsrByDst of the swapped
//                                             edge_index; weights permuted with TfgxPermuteRows)          tfgx.h:263-266
//   d/dw of TfgxSegmentReduce               = TfgxSddmm(g, x)                                              tfgx_sddmm_f32
//   d/dkernel, d/dbias of TfgxGemmBiasAct   = TfgxGemmTn(x, g)                                             tfgx_gemm_tn_f32
//   d/dx of TfgxGemmBiasAct                 = TfgxGemmBiasAct(g, kernel^T)
//   the ReLU of an epilogue                 = TfgxReluBackward(g, out)                                     tfgx_relu_backward_f32
// what tf.GradientTape derives for the reference's own composition (demo/demo_gcn.py:68-77).
// ---------------------------------------------------------------------------------------------------------------------
REGISTER_OP("TfgxSddmm")
    .Input("row_ptr: int32")
    .Input("col: int32")
    .Input("a: float")          // [N, F] rows indexed by destination (the upstream gradient)
    .Input("b: float")          // [N, F] rows indexed by source (the layer input)
    .Output("out: float");      // [E] <a[row(i)], b[col[i]]> per CSR position

class TfgxSddmmOp : public OpKernel {
 public:
  explicit TfgxSddmmOp(OpKernelConstruction* c) : OpKernel(c) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &a = ctx->input(2), &b = ctx->input(3);
    const int64_t n = rp.dim_size(0) - 1, E = col.dim_size(0), F = a.dim_size(1);
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {E}, &out));
    OP_REQUIRES(ctx, tfgx_sddmm_f32(rp.flat<int32>().data(), col.flat<int32>().data(), n, a.flat<float>().data(), F,
                                    b.flat<float>().data(), F, F, out->flat<float>().data(), TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
};
REGISTER_KERNEL_BUILDER(Name("TfgxSddmm").Device(DEVICE_GPU), TfgxSddmmOp);

REGISTER_OP("TfgxPermuteRows")
    .Input("src: float")        // [E] or [E, k] in the caller's edge order
    .Input("perm: int32")       // [E] CSR position -> caller's edge id (TfgxBuildCsrByDst)
    .Output("dst: float");      // the same rows in CSR order

class TfgxPermuteRowsOp : public OpKernel {
 public:
  explicit TfgxPermuteRowsOp(OpKernelConstruction* c) : OpKernel(c) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor &src = ctx->input(0), &perm = ctx->input(1);
    const int64_t E = perm.dim_size(0), width = src.dims() == 2 ? src.dim_size(1) : 1;
    Tensor* dst = nullptr;
    if (src.dims() == 2) OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {E, width}, &dst));
    else OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {E}, &dst));
    OP_REQUIRES(ctx, tfgx_permute_rows_f32(src.flat<float>().data(), perm.flat<int32>().data(), E, width,
                                           dst->flat<float>().data(), TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
};
REGISTER_KERNEL_BUILDER(Name("TfgxPermuteRows").Device(DEVICE_GPU), TfgxPermuteRowsOp);

REGISTER_OP("TfgxGemmTn")
    .Input("x: float")          // [M, K]   the layer input
    .Input("g: float")          // [M, N]   gradient of the layer output (after the ReLU mask)
    .Output("dw: float")        // [K, N] = x^T @ g
    .Output("db: float");       // [N]    = column sums of g

class TfgxGemmTnOp : public OpKernel {
 public:
  explicit TfgxGemmTnOp(OpKernelConstruction* c) : OpKernel(c) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor &x = ctx->input(0), &g = ctx->input(1);
    const int64_t M = x.dim_size(0), K = x.dim_size(1), N = g.dim_size(1);
    OP_REQUIRES(ctx, g.dim_size(0) == M, errors::InvalidArgument("x and g do not agree on M"));
    Tensor *dw = nullptr, *db = nullptr, ws;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {K, N}, &dw));
    OP_REQUIRES_OK(ctx, ctx->allocate_output(1, {N}, &db));
    const size_t ws_bytes = tfgx_gemm_tn_workspace_bytes(M, K, N, 1);
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_UINT8, {static_cast<int64_t>(ws_bytes)}, &ws));
    OP_REQUIRES(ctx, tfgx_gemm_tn_f32(x.flat<float>().data(), K, g.flat<float>().data(), N, M, K, N,
                                      dw->flat<float>().data(), N, db->flat<float>().data(), ws.flat<uint8>().data(),
                                      ws_bytes, TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
};
REGISTER_KERNEL_BUILDER(Name("TfgxGemmTn").Device(DEVICE_GPU), TfgxGemmTnOp);

REGISTER_OP("TfgxReluBackward")
    .Input("g: float")          // [M, N]
    .Input("out: float")        // [M, N] the forward output whose epilogue applied the ReLU
    .Output("gout: float");     // g where out > 0, else 0

class TfgxReluBackwardOp : public OpKernel {
 public:
  explicit TfgxReluBackwardOp(OpKernelConstruction* c) : OpKernel(c) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor &g = ctx->input(0), &o = ctx->input(1);
    const int64_t M = g.dim_size(0), N = g.dim_size(1);
    Tensor* gout = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {M, N}, &gout));
    OP_REQUIRES(ctx, tfgx_relu_backward_f32(g.flat<float>().data(), N, o.flat<float>().data(), N, M, N,
                                            gout->flat<float>().data(), N, TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
  }
};
REGISTER_KERNEL_BUILDER(Name("TfgxReluBackward").Device(DEVICE_GPU), TfgxReluBackwardOp);

// ---------------------------------------------------------------------------------------------------------------------
// The sharded path from a TensorFlow host (include/tfgx_dist.h; one process per GPU).  The ncclComm_t is created ONCE per
// process by the Python side through ctypes (tfgx_dist_unique_id on rank 0 -> the host's control channel -> tfgx_dist_comm_init)
// and handed to the ops as an int64 scalar in host memory; the exchange description (per-round, per-peer counts; dense
// starts) are host-memory int64 vectors, the packed send indices a device int32 vector — exactly the arrays
// tf_geometric_amd.dist.sharded.ShardedGraph builds (round_send_counts / round_recv_counts / round_send_dense /
// send_idx_packed).  TfgxHaloExchange returns the [own | halo] source table; its gradient is TfgxHaloReverse.
// (Both wait for every round before returning: overlap with the local-source pass needs the start / finish pair to be
// two ops with a control dependency in between — left to the host that schedules them.)
// ---------------------------------------------------------------------------------------------------------------------
#include "tfgx_dist.h"
#include <vector>

namespace {
struct HaloPlanHolder {            // one plan per op instance; rebuilt when the description changes
  tfgx_halo_plan* plan = nullptr;
  std::vector<int64_t> key;
  ~HaloPlanHolder() { tfgx_halo_plan_destroy(plan); }
  int Get(int32_t world, int32_t rank, int32_t rounds, const Tensor& sc, const Tensor& rc, const Tensor& ds,
          const int32_t* send_idx, tfgx_halo_plan** out) {
    const int64_t n = int64_t(world) * rounds;
    std::vector<int64_t> k;
    k.push_back(world); k.push_back(rank); k.push_back(rounds); k.push_back(reinterpret_cast<int64_t>(send_idx));
    for (int64_t i = 0; i < n; ++i) { k.push_back(sc.flat<int64_t>().data()[i]); k.push_back(rc.flat<int64_t>().data()[i]); k.push_back(ds.flat<int64_t>().data()[i]); }
    if (plan == nullptr || k != key) {
      tfgx_halo_plan_destroy(plan);
      plan = nullptr;
      const int rcode = tfgx_halo_plan_create(world, rank, rounds, sc.flat<int64_t>().data(), rc.flat<int64_t>().data(),
                                              ds.flat<int64_t>().data(), send_idx, &plan);
      if (rcode != 0) return rcode;
      key = k;
    }
    *out = plan;
    return 0;
  }
};
}  // namespace

REGISTER_OP("TfgxHaloExchange")
    .Input("x_own: float")            // [n_own, F] this rank's rows
    .Input("send_idx: int32")         // device: packed local row ids, round-major then peer-major
    .Input("send_counts: int64")      // host [rounds * world]
    .Input("recv_counts: int64")      // host [rounds * world]
    .Input("send_dense_start: int64") // host [rounds * world]: >= 0 contiguous own rows (no pack), -1 packed
    .Input("comm: int64")             // host scalar: the ncclComm_t of tfgx_dist_comm_init
    .Attr("world: int")
    .Attr("rank: int")
    .Attr("rounds: int")
    .Output("table: float");          // [n_own + rows_received, F] = [own | halo]

class TfgxHaloExchangeOp : public OpKernel {
 public:
  explicit TfgxHaloExchangeOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("world", &world_));
    OP_REQUIRES_OK(c, c->GetAttr("rank", &rank_));
    OP_REQUIRES_OK(c, c->GetAttr("rounds", &rounds_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &x = ctx->input(0), &idx = ctx->input(1), &sc = ctx->input(2), &rc = ctx->input(3), &ds = ctx->input(4);
    void* comm = reinterpret_cast<void*>(ctx->input(5).flat<int64_t>().data()[0]);
    const int64_t n_own = x.dim_size(0), F = x.dim_size(1);
    tfgx_halo_plan* plan = nullptr;
    OP_REQUIRES(ctx, holder_.Get(world_, rank_, rounds_, sc, rc, ds, idx.flat<int32>().data(), &plan) == 0,
                errors::InvalidArgument(tfgx_dist_last_error()));
    const int64_t n_halo = tfgx_halo_plan_rows_received(plan), n_pack = tfgx_halo_plan_rows_packed(plan);
    Tensor *table = nullptr, send_buf;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n_own + n_halo, F}, &table));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_FLOAT, {n_pack * F + 1}, &send_buf));
    float* t = table->flat<float>().data();
    // own rows first (the dense entries of the plan are sent straight from the table: ld == F), then the exchange
    OP_REQUIRES(ctx, tfgx_gather_rows_f32(x.flat<float>().data(), F, nullptr /* identity */, n_own, F, t, F,
                                          TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
    OP_REQUIRES(ctx, tfgx_halo_exchange_start(plan, t, F, F, t + n_own * F, F, send_buf.flat<float>().data(),
                                              static_cast<size_t>(n_pack * F + 1), comm, TfStream(ctx), CommStream()) == 0,
                errors::Internal(tfgx_dist_last_error()));
    OP_REQUIRES(ctx, tfgx_halo_exchange_finish(plan, -1, TfStream(ctx)) == 0, errors::Internal(tfgx_dist_last_error()));
  }
  static void* CommStream() { return nullptr; }   // the real shim creates ONE hipStream_t per process here (second HIP stream)
  int world_, rank_, rounds_;
  HaloPlanHolder holder_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxHaloExchange").Device(DEVICE_GPU).HostMemory("send_counts").HostMemory("recv_counts")
                            .HostMemory("send_dense_start").HostMemory("comm"), TfgxHaloExchangeOp);

REGISTER_OP("TfgxHaloReverse")
    .Input("d_table: float")          // [n_own + rows_received, F] gradient w.r.t. the [own | halo] table
    .Input("send_idx: int32")
    .Input("send_counts: int64")
    .Input("recv_counts: int64")
    .Input("send_dense_start: int64")
    .Input("comm: int64")
    .Attr("world: int")
    .Attr("rank: int")
    .Attr("rounds: int")
    .Attr("n_own: int")
    .Output("d_own: float");          // [n_own, F]: own part + what the peers computed for this rank's rows

class TfgxHaloReverseOp : public OpKernel {
 public:
  explicit TfgxHaloReverseOp(OpKernelConstruction* c) : OpKernel(c) {
    OP_REQUIRES_OK(c, c->GetAttr("world", &world_));
    OP_REQUIRES_OK(c, c->GetAttr("rank", &rank_));
    OP_REQUIRES_OK(c, c->GetAttr("rounds", &rounds_));
    OP_REQUIRES_OK(c, c->GetAttr("n_own", &n_own_));
  }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &dt = ctx->input(0), &idx = ctx->input(1), &sc = ctx->input(2), &rc = ctx->input(3), &ds = ctx->input(4);
    void* comm = reinterpret_cast<void*>(ctx->input(5).flat<int64_t>().data()[0]);
    const int64_t F = dt.dim_size(1);
    tfgx_halo_plan* plan = nullptr;
    OP_REQUIRES(ctx, holder_.Get(world_, rank_, rounds_, sc, rc, ds, idx.flat<int32>().data(), &plan) == 0,
                errors::InvalidArgument(tfgx_dist_last_error()));
    const int64_t n_sent = tfgx_halo_plan_rows_sent(plan);
    Tensor *d_own = nullptr, back;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n_own_, F}, &d_own));
    OP_REQUIRES_OK(ctx, ctx->allocate_temp(DT_FLOAT, {n_sent * F + 1}, &back));
    const float* d = dt.flat<float>().data();
    OP_REQUIRES(ctx, tfgx_gather_rows_f32(d, F, nullptr /* identity */, n_own_, F, d_own->flat<float>().data(), F,
                                          TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
    OP_REQUIRES(ctx, tfgx_halo_reverse_start(plan, d + n_own_ * F, F, back.flat<float>().data(),
                                             static_cast<size_t>(n_sent * F + 1), comm, TfStream(ctx),
                                             TfgxHaloExchangeOp::CommStream()) == 0,
                errors::Internal(tfgx_dist_last_error()));
    OP_REQUIRES(ctx, tfgx_halo_reverse_finish(plan, d_own->flat<float>().data(), F, F, back.flat<float>().data(),
                                              TfStream(ctx)) == 0,
                errors::Internal(tfgx_dist_last_error()));
  }
  int world_, rank_, rounds_;
  int64_t n_own_;
  HaloPlanHolder holder_;
};
REGISTER_KERNEL_BUILDER(Name("TfgxHaloReverse").Device(DEVICE_GPU).HostMemory("send_counts").HostMemory("recv_counts")
                            .HostMemory("send_dense_start").HostMemory("comm"), TfgxHaloReverseOp);

REGISTER_OP("TfgxAllReduceSum")
    .Input("x: float")
    .Input("comm: int64")
    .Output("out: float");            // sum over the ranks (weight gradients of replicated layers; demo_distributed_gcn.py:52-57)

class TfgxAllReduceSumOp : public OpKernel {
 public:
  explicit TfgxAllReduceSumOp(OpKernelConstruction* c) : OpKernel(c) {}
  void Compute(OpKernelContext* ctx) override {
    const Tensor& x = ctx->input(0);
    void* comm = reinterpret_cast<void*>(ctx->input(1).flat<int64_t>().data()[0]);
    const int64_t n = x.NumElements();
    Tensor* out = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, {n}, &out));
    OP_REQUIRES(ctx, tfgx_gather_rows_f32(x.flat<float>().data(), n, nullptr /* identity */, 1, n,
                                          out->flat<float>().data(), n, TfStream(ctx)) == 0,
                errors::Internal(tfgx_last_error()));
    OP_REQUIRES(ctx, tfgx_allreduce_sum_f32(out->flat<float>().data(), n, comm, TfStream(ctx)) == 0,
                errors::Internal(tfgx_dist_last_error()));
  }
};
REGISTER_KERNEL_BUILDER(Name("TfgxAllReduceSum").Device(DEVICE_GPU).HostMemory("comm"), TfgxAllReduceSumOp);
