# coding=utf-8
"""Next-row-2 convolutions (GIN / SGC / TAGCN / APPNP / SSGC / ChebyNet / LEConv): compositions of the hot-path
kernels vs the oracle's line-by-line restatement."""
import numpy as np
import pytest
import torch

from conftest import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(oracle):
    n, f = 400, 14
    ei = oracle.synthetic_edges(n, 4000, seed=21)
    ei = np.concatenate([ei, np.stack([np.arange(5, dtype=np.int32)] * 2)], axis=1)    # a few explicit self-loops
    rng = np.random.Generator(np.random.PCG64(22))
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32)
    return dict(n=n, f=f, ei=ei, x=x, w=w, rng=rng)


def _k(oracle, rng, a, b):
    return oracle.glorot_uniform(rng, a, b)


def test_sgc_tagcn(tfg, oracle, g):
    rng = g["rng"]
    kernel, bias = _k(oracle, rng, g["f"], 9), (rng.standard_normal(9) * 0.1).astype(np.float32)
    for k in (1, 3):
        layer = tfg.layers.SGC(9, k=k, activation=tfg.relu)
        layer._maybe_build([g["x"]])
        layer.set_weights(kernel=kernel, bias=bias)
        assert_parity(layer([g["x"], g["ei"], g["w"]], cache={}).cpu().numpy(),
                      oracle.sgc(g["x"], g["ei"], g["w"], k, kernel, bias, "relu"), what="SGC k={}".format(k))
    tk = _k(oracle, rng, g["f"] * 4, 7)
    layer = tfg.layers.TAGCN(7, k=3)
    layer._maybe_build([g["x"]])
    layer.set_weights(kernel=tk)
    assert_parity(layer([g["x"], g["ei"], g["w"]]).cpu().numpy(),
                  oracle.tagcn(g["x"], g["ei"], g["w"], 3, tk, np.zeros(7, np.float32)), what="TAGCN")


def test_appnp_ssgc(tfg, oracle, g):
    rng = g["rng"]
    ks = [_k(oracle, rng, g["f"], 16), _k(oracle, rng, 16, 6)]
    bs = [(rng.standard_normal(16) * 0.1).astype(np.float32), (rng.standard_normal(6) * 0.1).astype(np.float32)]
    appnp = tfg.layers.APPNP([16, 6], k=6, alpha=0.15)
    appnp._maybe_build([g["x"]])
    appnp.set_weights(kernel_0=ks[0], bias_0=bs[0], kernel_1=ks[1], bias_1=bs[1])
    assert_parity(appnp([g["x"], g["ei"], g["w"]], cache={}).cpu().numpy(),
                  oracle.appnp(g["x"], g["ei"], g["w"], ks, bs, "relu", None, k=6, alpha=0.15), what="APPNP")
    ssgc = tfg.layers.SSGC([16, 6], k=5, alpha=0.2)
    ssgc._maybe_build([g["x"]])
    ssgc.set_weights(kernel_0=ks[0], bias_0=bs[0], kernel_1=ks[1], bias_1=bs[1])
    assert_parity(ssgc([g["x"], g["ei"], g["w"]]).cpu().numpy(),
                  oracle.ssgc(g["x"], g["ei"], g["w"], ks, bs, k=5, alpha=0.2), what="SSGC")
    plain = tfg.layers.SSGC(None, k=4)
    assert_parity(plain([g["x"], g["ei"]]).cpu().numpy(), oracle.ssgc(g["x"], g["ei"], None, None, None, k=4),
                  what="SSGC without MLP")


@pytest.mark.parametrize("norm", ["sym", "rw", None])
@pytest.mark.parametrize("k", [1, 2, 4])
def test_chebynet(tfg, oracle, g, norm, k):
    rng = np.random.Generator(np.random.PCG64(k))
    kernels = [_k(oracle, rng, g["f"], 5) for _ in range(k)]
    bias = (rng.standard_normal(5) * 0.1).astype(np.float32)
    layer = tfg.layers.ChebyNet(5, k, activation=tfg.relu, normalization_type=norm)
    layer._maybe_build([g["x"]])
    layer.set_weights(bias=bias, **{"kernel{}".format(i): kernels[i] for i in range(k)})
    got = layer([g["x"], g["ei"], g["w"]], cache={}).cpu().numpy()
    ref = oracle.chebynet(g["x"], g["ei"], g["w"], k, kernels, bias, "relu", norm)
    tol = 1e-5 if norm is not None else 1e-4        # the un-normalised variant works with degree-sized values
    assert_parity(got, ref, tol=tol, what="ChebyNet {} k={}".format(norm, k))


def test_gin_leconv(tfg, oracle, g):
    rng = g["rng"]
    w1 = _k(oracle, rng, g["f"], 8)
    mlp_t = lambda h, training=None: torch.relu(h @ tfg._lib.as_f32(w1))
    mlp_o = lambda h: np.maximum(oracle.matmul(h, w1), 0)
    got = tfg.layers.GIN(mlp_t, eps=0.3)([g["x"], g["ei"]]).cpu().numpy()
    assert_parity(got, mlp_o(oracle.gin(g["x"], g["ei"], lambda h: h, eps=0.3)), tol=2e-5, what="GIN")
    ks = [_k(oracle, rng, g["f"], 6) for _ in range(3)]
    b0, b1 = (rng.standard_normal(6) * 0.1).astype(np.float32), (rng.standard_normal(6) * 0.1).astype(np.float32)
    layer = tfg.layers.LEConv(6, activation=tfg.relu)
    layer._maybe_build([g["x"]])
    layer.set_weights(self_kernel=ks[0], self_bias=b0, aggr_self_kernel=ks[1], aggr_self_bias=b1, aggr_neighbor_kernel=ks[2])
    assert_parity(layer([g["x"], g["ei"], g["w"]]).cpu().numpy(),
                  oracle.le_conv(g["x"], g["ei"], g["w"], ks[0], b0, ks[1], b1, ks[2], None, "relu"), what="LEConv")


def test_propagation_layers_are_trainable(tfg, g):
    layer = tfg.layers.APPNP([12, 4], k=3)
    layer._maybe_build([g["x"]])
    layer.trainable(True)
    out = layer([g["x"], g["ei"], g["w"]], cache={})
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0
               for p in layer.parameters())


def test_sparse_node_features_across_convs(tfg, oracle, g):
    """Sparse x (the reference's isinstance(x, tf.sparse.SparseTensor) branches: gat.py:47-68, sgc.py:31, tagcn.py:32,
    appnp.py:64, ssgc.py:73, chebynet.py:100-119): same outputs as the dense features they encode, as a SparseMatrix and
    as a torch sparse COO tensor; GAT's kernels receive gradients through the sparse projections."""
    rng = np.random.Generator(np.random.PCG64(77))
    n, f = g["n"], g["f"]
    dense = ((rng.random((n, f)) < 0.2) * rng.standard_normal((n, f))).astype(np.float32)
    dense[3] = 0
    r, c = np.nonzero(dense)
    xs = tfg.SparseMatrix(np.stack([r, c]).astype(np.int32), dense[r, c], [n, f])
    xt = torch.sparse_coo_tensor(np.stack([r, c]), dense[r, c], (n, f)).cuda()
    ei, w = g["ei"], g["w"]

    def both(make, call, what, tol=1e-5):
        layer = make()
        ref = call(layer, dense).cpu().numpy()
        assert_parity(call(layer, xs).cpu().numpy(), ref, tol=tol, what=what + " (SparseMatrix)")
        assert_parity(call(layer, xt).cpu().numpy(), ref, tol=tol, what=what + " (torch COO)")

    both(lambda: tfg.layers.GAT(16, attention_units=8, num_heads=4, activation=tfg.relu),
         lambda l, x: l([x, ei]), "GAT", tol=2e-5)
    both(lambda: tfg.layers.SGC(9, k=2, activation=tfg.relu), lambda l, x: l([x, ei, w], cache={}), "SGC")
    both(lambda: tfg.layers.TAGCN(7, k=2), lambda l, x: l([x, ei, w]), "TAGCN")
    both(lambda: tfg.layers.APPNP([16, 6], k=4, alpha=0.1), lambda l, x: l([x, ei, w]), "APPNP")
    both(lambda: tfg.layers.SSGC([16, 6], k=3, alpha=0.2), lambda l, x: l([x, ei, w]), "SSGC")
    both(lambda: tfg.layers.SSGC(None, k=3), lambda l, x: l([x, ei]), "SSGC without MLP")
    both(lambda: tfg.layers.ChebyNet(8, k=3), lambda l, x: l([x, ei, w]), "ChebyNet")
    gat = tfg.layers.GAT(8, attention_units=4, num_heads=2)
    gat.trainable(True)
    gat([xs, ei]).square().sum().backward()
    gs = {k: getattr(gat, k).grad.clone() for k in ("query_kernel", "key_kernel", "kernel")}
    for p_ in gat.parameters():
        p_.grad = None
    gat([dense, ei]).square().sum().backward()
    for k, v in gs.items():
        assert_parity(v.cpu().numpy(), getattr(gat, k).grad.cpu().numpy(), tol=1e-4, what="GAT d/d" + k + " via sparse x")
