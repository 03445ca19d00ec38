# coding=utf-8
"""Tiny stand-in for the tf.keras.Model plumbing the reference layers rely on (add_weight / build / call)."""
import math

import torch

from .. import _lib as L


class Layer(object):
    """Weights are created lazily from the first input's feature width, as Keras `build` does
    (e.g. layers/conv/gcn.py:17-30).  `layer(inputs, **kw)` == `layer.call(inputs, **kw)`."""

    def __init__(self, seed=None, name=None):
        self.built = False
        self.name = name or self.__class__.__name__
        self._weights = {}
        self._init_of = {}
        self._attr_of = {}
        self._gen = None
        self._seed = seed
        self._trainable = False

    def _rng(self, dev):
        if self._gen is None:
            self._gen = torch.Generator(device="cpu")
            self._gen.manual_seed(0 if self._seed is None else int(self._seed))
        return self._gen

    def add_weight(self, name, shape, initializer="glorot_uniform"):
        dev = L.device()
        if initializer == "zeros":
            w = torch.zeros(shape, dtype=torch.float32, device=dev)
        elif initializer == "glorot_uniform":
            fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            w = ((torch.rand(shape, generator=self._rng(dev), dtype=torch.float32) * 2.0 - 1.0) * limit).to(dev)
        else:
            raise ValueError("unknown initializer {}".format(initializer))
        self._weights[name] = w
        self._init_of[name] = initializer
        return w

    def set_weights(self, **named):
        """Load weights by the reference's variable names (checkpoint compatibility)."""
        for k, v in named.items():
            # `k` is the VARIABLE name the reference registered with add_weight (what a checkpoint stores); the python
            # attribute may differ (MaxPoolGraphSage: variable "mlp_kernel" lives in self.neighbor_mlp_kernel,
            # layers/conv/graph_sage.py:322-328).  Attribute names are accepted too.
            attr = self._attr_of.get(k)
            if attr is None and k in self._weights and self._weights[k] is not None:
                old = self._weights[k]
                attr = next((a for a, val in self.__dict__.items() if val is old and a != "_weights"), None)
            if attr is None and hasattr(self, k):
                attr = k
            if attr is None:
                raise KeyError("{} has no weight named {}".format(self.name, k))
            self._attr_of[k] = attr              # variable name -> python attribute, resolved once
            # the layer OWNS its weights: a caller's float32 device tensor is copied, never aliased (an optimizer step
            # or requires_grad_ must not write into user data)
            t = L.as_f32(v).detach().clone().contiguous()
            if self._trainable:
                t.requires_grad_(True)
            name = k if k in self._weights else next((n for n, val in self._weights.items()
                                                      if val is getattr(self, attr)), k)
            setattr(self, attr, t)
            self._weights[name] = t

    @property
    def weights(self):
        return dict(self._weights)

    @property
    def losses(self):
        """Regularisation terms, as keras collects them in `layer.losses`: the reference hands `kernel_regularizer` to every
        glorot-initialised matrix and `bias_regularizer` to every zero-initialised vector it registers (e.g.
        layers/conv/gcn.py:26-30, layers/conv/gat.py:64-83, layers/conv/graph_sage.py:55-61); scalars (GIN's eps,
        layers/conv/gin.py:23) take none.  One term per regularised weight, evaluated on the CURRENT weights; differentiable
        when the layer is trainable."""
        out = []
        for name, w in self._weights.items():
            if w is None or w.dim() == 0:
                continue
            kind = "bias_regularizer" if self._init_of.get(name) == "zeros" else "kernel_regularizer"
            reg = getattr(self, kind, None)
            if reg is not None:
                out.append(reg(w))
        return out

    def parameters(self):
        """The layer's weight tensors (for torch.optim); call trainable(True) first to track gradients."""
        return [t for t in self._weights.values() if t is not None]

    def trainable(self, flag=True):
        """Turn gradient tracking of every weight on/off.  With it on, calls run through the differentiable
        kernels of tf_geometric_amd.autograd (the role tf.GradientTape plays for the reference's keras layers)."""
        self._trainable = bool(flag)          # also applies to weights created later (lazy build on the first call)
        for t in self._weights.values():
            if t is not None:
                t.requires_grad_(flag)
        return self

    def build(self, input_shapes):
        raise NotImplementedError

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def _maybe_build(self, inputs):
        if not self.built:
            L.require_gpu()
            x = inputs[0]
            self.build([tuple(x.shape)])
            self.built = True
            if self._trainable:
                self.trainable(True)

    def __call__(self, inputs, **kwargs):
        self._maybe_build(inputs)
        return self.call(inputs, **kwargs)
