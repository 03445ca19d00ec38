# coding=utf-8
"""The `tf.load_op_library` binding (integration/tf_shim/tfgx_tf_ops.cc) EXECUTED: linked against the mock TensorFlow runtime
of integration/tf_shim/mock/ (real device tensors, op / kernel registries, attr defaults, host-memory pinning) and
libtfgx.so / libtfgx_dist.so, every op's Compute() body — shape checks, argument marshalling into the C ABI — runs on the GPU
through lib/tf_shim_mock_driver.  The pipelines are the ones integration/tf_shim/tfgx_tf.py composes for a
tfg.layers.GCN([x, edge_index, edge_weight]) call (layers/conv/gcn.py:129-156); outputs are held against the REFERENCE'S OWN
outputs (tests/golden/reference_cases.npz, case "gcn").  Executed against a mock runtime; never against TensorFlow (not
installable in this image)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_parity
import reference_cases as rc

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "tf_geometric_amd", "lib", "tf_shim_mock_driver")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_cases.npz")
_DT = {"float32": "float", "int32": "int32", "int64": "int64"}


class Script(object):
    def __init__(self, workdir):
        self.dir, self.lines, self.saved = str(workdir), [], []

    def input(self, name, arr, host=False):
        arr = np.ascontiguousarray(arr)
        arr.tofile(os.path.join(self.dir, name + ".bin"))
        dims = ",".join(str(d) for d in arr.shape) if arr.ndim else "scalar"
        self.lines.append("input {} {} {} {} {}.bin".format(name, _DT[str(arr.dtype)], "host" if host else "device", dims, name))

    def op(self, name, ins, outs, expect_error=None, **attrs):
        a = " ".join("attr={}:{}:{}".format(k, "bool" if isinstance(v, bool) else "float" if isinstance(v, float) else "int",
                                            str(v).lower() if isinstance(v, bool) else v) for k, v in attrs.items())
        line = "op {} in={} out={} {}".format(name, ",".join(ins), ",".join(outs), a).rstrip()
        self.lines.append(line if expect_error is None else "expect_error {} {}".format(expect_error, line))

    def save(self, name, dtype=np.float32):
        self.lines.append("save {} {}.out".format(name, name))
        self.saved.append((name, dtype))

    def run(self):
        path = os.path.join(self.dir, "script.txt")
        with open(path, "w") as fh:
            fh.write("\n".join(self.lines) + "\n")
        res = subprocess.run([DRIVER, path, self.dir], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        text = res.stdout.decode()
        assert res.returncode == 0 and "TF_SHIM_MOCK_OK" in text, text[-3000:]
        out = {}
        for ln in text.splitlines():
            if ln.startswith("saved "):
                _, name, _, dims = ln.split()
                shape = tuple(int(d) for d in dims[5:].split(",")) if dims[5:] else ()
                dtype = dict(self.saved)[name]
                out[name] = np.fromfile(os.path.join(self.dir, name + ".out"), dtype=dtype).reshape(shape)
        return out, text


def _gcn_case():
    case = [c for c in rc.CASES if c.name == "gcn"][0]
    return case.inputs(), dict(np.load(GOLDEN))


NORM = {"both": 0, "left": 1, "right": 2}
CFGS = [dict(), dict(renorm=False), dict(improved=True), dict(renorm=False, improved=True), dict(norm="left"),
        dict(norm="right"), dict(add_self_loop=False), dict(norm="left", add_self_loop=False),
        dict(norm="right", add_self_loop=False)]          # (sym=False needs column degrees: not in the op's signature)


def test_gcn_layer_through_the_shim_ops_matches_the_reference(tfg, tmp_path):
    assert os.path.exists(DRIVER), "run __graft_entry__.build()"
    g, golden = _gcn_case()
    n, f = g["n"], g["x"].shape[1]
    s = Script(tmp_path)
    s.input("ei", g["ei"].astype(np.int32))
    s.input("w", g["w"].astype(np.float32))
    s.input("x", g["x"].astype(np.float32))
    for k in ("kernel", "bias", "wide_kernel", "wide_bias"):
        s.input(k, g[k].astype(np.float32))
    s.input("none", np.zeros(0, np.float32))
    s.op("TfgxBuildCsrByDst", ["ei"], ["row_ptr", "col", "perm"], num_nodes=n)
    s.op("TfgxPermuteRows", ["w", "perm"], ["w_csr"])
    s.op("TfgxGemmBiasAct", ["x", "kernel", "none"], ["h"])                      # x @ kernel (gcn.py:272), defaults: act = 0
    for i, cfg in enumerate(CFGS):
        s.op("TfgxGcnNormEdges", ["row_ptr", "col", "w_csr"], ["wn%d" % i, "sc%d" % i], norm=NORM[cfg.get("norm", "both")],
             add_self_loop=cfg.get("add_self_loop", True), renorm=cfg.get("renorm", True), improved=cfg.get("improved", False))
        s.op("TfgxSegmentReduce", ["row_ptr", "col", "wn%d" % i, "h", "sc%d" % i, "bias"], ["o%d" % i], op=0, act=1)
        s.save("o%d" % i)
    # units > F: the aggregate-then-project route in ONE op, and as two ops
    s.op("TfgxAggregateGemm", ["row_ptr", "col", "wn0", "x", "sc0", "wide_kernel", "wide_bias"], ["wide_fused", "no_agg"], op=0, act=1)
    s.op("TfgxAggregateGemm", ["row_ptr", "col", "wn0", "x", "sc0", "wide_kernel", "wide_bias"], ["wide_fused_t", "wide_agg"],
         op=0, act=1, want_aggregate=True)                 # the training form: the aggregate as the second output
    s.op("TfgxSegmentReduce", ["row_ptr", "col", "wn0", "x", "sc0", "none"], ["agg"], op=0)
    s.op("TfgxGemmBiasAct", ["agg", "wide_kernel", "wide_bias"], ["wide_two"], act=1)
    s.op("TfgxSegmentReduce", ["row_ptr", "col", "wn0", "h", "sc0", "none"], ["plain"], op=0)      # no bias, no activation
    s.op("TfgxGcnNormEdges", ["row_ptr", "col", "none"], ["wn_u", "sc_u"])                          # unweighted graph (w = ones)
    s.op("TfgxSegmentReduce", ["row_ptr", "col", "wn_u", "h", "sc_u", "none"], ["unweighted"], op=0)
    for name in ("wide_fused", "wide_two", "plain", "unweighted", "wide_fused_t", "wide_agg", "agg", "no_agg"):
        s.save(name)
    s.save("row_ptr", np.int32)
    s.save("col", np.int32)
    s.save("perm", np.int32)
    out, text = s.run()
    for i, cfg in enumerate(CFGS):
        key = "gcn::" + (",".join("{}={}".format(k, v) for k, v in sorted(cfg.items())) or "default")
        assert_parity(out["o%d" % i], golden[key], what="shim ops vs reference " + key)
    assert_parity(out["wide_fused"], golden["gcn::wide"], what="TfgxAggregateGemm vs reference gcn::wide")
    assert_parity(out["wide_two"], golden["gcn::wide"], what="TfgxSegmentReduce + TfgxGemmBiasAct vs reference gcn::wide")
    assert np.array_equal(out["wide_fused_t"], out["wide_fused"]) and np.array_equal(out["wide_agg"], out["agg"])
    assert out["no_agg"].size == 0 and out["wide_agg"].shape == (n, f)
    assert_parity(out["plain"], golden["gcn::no_bias_no_act"], what="shim ops vs reference gcn::no_bias_no_act")
    assert_parity(out["unweighted"], golden["gcn::unweighted"], what="shim ops vs reference gcn::unweighted")
    # the plan the op built: stable sort of the edges by destination, bit for bit
    order = np.argsort(g["ei"][0], kind="stable")
    assert np.array_equal(out["perm"], order.astype(np.int32)) and np.array_equal(out["col"], g["ei"][1][order])
    assert np.array_equal(out["row_ptr"], np.concatenate([[0], np.cumsum(np.bincount(g["ei"][0], minlength=n))]).astype(np.int32))


def test_backward_gat_and_sharded_ops_through_the_shim(tfg, tmp_path):
    """The backward ops (TfgxGemmTn / TfgxReluBackward / TfgxSddmm), TfgxGatFused and the sharded-path ops on a 1-rank
    communicator (no peer: the table is the own rows, the reverse returns the own part, the all-reduce is the identity)."""
    import torch
    from tf_geometric_amd.nn.conv.gat import gat_attention
    from tf_geometric_amd.plan import CsrPlan
    g, _ = _gcn_case()
    n = g["n"]
    rng = np.random.Generator(np.random.PCG64(5))
    x = g["x"].astype(np.float32)
    f = x.shape[1]
    gout = rng.standard_normal((n, 10)).astype(np.float32)
    fwd = np.maximum(rng.standard_normal((n, 10)), 0).astype(np.float32)
    Q, K, V = (rng.standard_normal((n, 8)).astype(np.float32) for _ in range(3))
    s = Script(tmp_path)
    s.input("ei", g["ei"].astype(np.int32))
    s.input("x", x)
    s.input("gout", gout)
    s.input("fwd", fwd)
    for name, arr in (("Q", Q), ("K", K), ("V", V)):
        s.input(name, arr)
    s.op("TfgxBuildCsrByDst", ["ei"], ["row_ptr", "col", "perm"], num_nodes=n)
    s.op("TfgxReluBackward", ["gout", "fwd"], ["gm"])
    s.op("TfgxGemmTn", ["x", "gm"], ["dw", "db"])
    s.input("b10", rng.standard_normal((n, 10)).astype(np.float32))
    s.op("TfgxSddmm", ["row_ptr", "col", "gm", "b10"], ["dwe"])
    s.op("TfgxGatFused", ["row_ptr", "col", "Q", "K", "V"], ["att"], num_heads=2)
    # sharded path, world = 1
    s.lines.append("comm comm")
    s.input("idx0", np.zeros(0, np.int32))
    s.input("zero1", np.zeros(1, np.int64), host=True)
    s.input("minus1", np.full(1, -1, np.int64), host=True)
    s.op("TfgxHaloExchange", ["x", "idx0", "zero1", "zero1", "minus1", "comm"], ["table"], world=1, rank=0, rounds=1)
    s.op("TfgxHaloReverse", ["gout", "idx0", "zero1", "zero1", "minus1", "comm"], ["d_own"], world=1, rank=0, rounds=1, n_own=n)
    s.input("flat", x.reshape(-1))
    s.op("TfgxAllReduceSum", ["flat", "comm"], ["summed"])
    for name in ("gm", "dw", "db", "dwe", "att", "table", "d_own", "summed"):
        s.save(name)
    s.save("row_ptr", np.int32)
    s.save("col", np.int32)
    # the shim's own argument checks and the C ABI's error strings surface as op errors
    s.input("ei3", np.zeros((3, 4), np.int32))
    s.op("TfgxBuildCsrByDst", ["ei3"], ["a", "b", "c"], expect_error="edge_index", num_nodes=n)
    s.input("ei_bad", np.array([[0, n + 5], [1, 2]], np.int32))
    s.op("TfgxBuildCsrByDst", ["ei_bad"], ["a", "b", "c"], expect_error="InvalidArgument", num_nodes=n)
    s.input("k_bad", np.zeros((f + 1, 4), np.float32))
    s.input("none", np.zeros(0, np.float32))
    s.op("TfgxGemmBiasAct", ["x", "k_bad", "none"], ["a"], expect_error="agree")
    s.op("TfgxGemmBiasAct", ["x", "k_bad"], ["a"], expect_error="inputs")                         # registration: 3 inputs
    s.op("TfgxSegmentReduce", ["row_ptr", "col", "none", "x", "none", "none"], ["a"], expect_error="op")     # attr without default
    s.op("TfgxHaloExchange", ["x", "idx0", "idx0", "zero1", "minus1", "comm"], ["a"], expect_error="dtype", world=1, rank=0, rounds=1)
    out, text = s.run()
    assert text.count("expected_error") == 6, text
    gm = np.where(fwd > 0, gout, 0).astype(np.float32)
    assert np.array_equal(out["gm"], gm)
    assert_parity(out["dw"], x.astype(np.float64).T @ gm.astype(np.float64), tol=1e-4, what="TfgxGemmTn dW")
    assert_parity(out["db"], gm.astype(np.float64).sum(0), tol=1e-4, what="TfgxGemmTn db")
    rows = np.repeat(np.arange(n), np.diff(out["row_ptr"]))
    b10 = np.fromfile(os.path.join(str(tmp_path), "b10.bin"), np.float32).reshape(n, 10)
    assert_parity(out["dwe"], (gm[rows].astype(np.float64) * b10[out["col"]]).sum(1), tol=1e-4, what="TfgxSddmm")
    plan = CsrPlan.build(tfg._lib.as_i32(g["ei"]), n, n)
    ref = gat_attention(plan, tfg._lib.as_f32(Q), tfg._lib.as_f32(K), tfg._lib.as_f32(V), 2)
    assert torch.equal(torch.from_numpy(out["att"]), ref.cpu())            # same kernel, same arguments: bit-identical
    assert np.array_equal(out["table"], x) and np.array_equal(out["d_own"], gout) and np.array_equal(out["summed"], x.reshape(-1))
