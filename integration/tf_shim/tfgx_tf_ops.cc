// TensorFlow custom-op shim over the tfgx C ABI (include/tfgx.h) — the `tf.load_op_library` route BASELINE.json's
// north_star names.  NOT BUILT OR TESTED IN THIS REPO: TensorFlow (and tf_sparse) are absent from the image and
// cannot be installed (no network), so this file is the reference-side binding a maintainer with a TensorFlow-ROCm
// build would compile (build_tf_shim.sh) and load; the executable, tested boundary here is the ctypes/PyTorch host
// in tf_geometric_amd/.  Each op forwards raw device pointers + sizes to one C-ABI call on TF's own HIP stream.
//
//   TfgxBuildCsrByDst   edge_index[2,E] -> row_ptr[N+1], col[E], perm[E]          (tfgx_build_csr_by_dst)
//   TfgxSegmentReduce   plan + x[N,F] (+ w[E] in CSR order) -> out[N,F]           (tfgx_segment_reduce_f32)
//                       replaces tf.gather + gcn_mapper + tf.math.unsorted_segment_{sum,mean,max}
//                       (tf_geometric/nn/kernel/map_reduce.py:15-42,60-70)
//   TfgxGatFused        plan + Q,K,V -> out                                       (tfgx_gat_fused_f32)
//                       replaces tf_geometric/nn/conv/gat.py:56-89,112
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"
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
    .Attr("op: int")            // 0 sum, 1 mean, 2 max
    .Output("out: float");

class TfgxSegmentReduceOp : public OpKernel {
 public:
  explicit TfgxSegmentReduceOp(OpKernelConstruction* c) : OpKernel(c) { OP_REQUIRES_OK(c, c->GetAttr("op", &op_)); }
  void Compute(OpKernelContext* ctx) override {
    const Tensor &rp = ctx->input(0), &col = ctx->input(1), &w = ctx->input(2), &x = ctx->input(3);
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
    OP_REQUIRES(ctx, tfgx_segment_reduce_f32(&a, TfStream(ctx)) == 0, errors::Internal(tfgx_last_error()));
  }
  int op_;
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
