# coding=utf-8
"""Activation handling: None / "relu" / relu callables are fused into kernel epilogues; any other callable runs
after the kernel on the torch tensor (the reference applies `activation(h)` last, e.g. nn/conv/gcn.py:287-288)."""
import torch

from . import _lib as L


def relu(x):
    """Stand-in for tf.nn.relu (the reference's default activation for GraphSAGE / GAT Q,K:
    layers/conv/graph_sage.py:13, layers/conv/gat.py:15-16)."""
    return torch.relu(x)


_RELU_FNS = {relu, torch.relu, torch.nn.functional.relu}


def resolve(activation):
    """-> (fused act code, python callable applied afterwards or None)."""
    if activation is None:
        return L.ACT_NONE, None
    if isinstance(activation, str):
        if activation == "relu":
            return L.ACT_RELU, None
        if activation in ("linear", "none"):
            return L.ACT_NONE, None
        raise ValueError("unknown activation {!r}".format(activation))
    if activation in _RELU_FNS or isinstance(activation, torch.nn.ReLU):
        return L.ACT_RELU, None
    if callable(activation):
        return L.ACT_NONE, activation
    raise ValueError("activation must be None, 'relu' or a callable")
