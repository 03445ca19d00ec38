# coding=utf-8
"""HIP-graph replay of a whole forward pass.

The reference hides per-op launch cost behind tf.function (a tracing compiler; "10X faster" in its own words,
demo/demo_gcn.py:107-109).  Here every operator is already one or two kernel launches on the current HIP stream
with caller-owned buffers, so a model's forward over a CACHED plan is a fixed launch sequence: it is captured once
into a hipGraph and replayed — no tracing, no compiler.  On small graphs (ogbn-arxiv shape: a 2-layer GCN is six
launches of 50-150 us) the replay removes the host-side cost of issuing them.

Capture uses torch.cuda.CUDAGraph purely as the capture/replay plumbing (hipStreamBeginCapture / hipGraphLaunch and
a capture-aware allocator); the nodes of the graph are this library's kernels.
"""
import torch

from . import _lib as L


class CapturedForward(object):
    """fn(*tensors) -> tensor, captured once.  Plans / normalised adjacencies must already be cached (building a
    plan synchronises and cannot be captured) — pass the same `cache` dict the eager call used."""

    def __init__(self, fn, *example_inputs, warmup=2):
        L.require_gpu()
        self.static_inputs = [L.as_f32(t).clone() if isinstance(t, torch.Tensor) or hasattr(t, "shape") else t
                              for t in example_inputs]
        from .plan import no_auto_promotion
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # the static input buffers are rewritten before every replay: the warm-up's repeated calls on them must not be read as
        # "the same features again" (plan.static_rows' automatic promotion); tensors `fn` closes over are covered by the rule
        # that a captured sequence only ever uses a layout the caller DECLARED (prepare_static_features)
        with torch.cuda.stream(side), torch.no_grad(), no_auto_promotion():
            for _ in range(warmup):                # builds / fetches cached plans, sizes the allocator pools
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if isinstance(dst, torch.Tensor):
                dst.copy_(L.as_f32(src), non_blocking=True)
        self.graph.replay()
        return self.static_output


class CapturedTrainStep(object):
    """One whole training step — zero the gradients, forward, loss, backward through the kernels' own backward
    (tf_geometric_amd.autograd), optimizer update — captured once into a hipGraph and replayed.

    The reference wraps its forward in tf.function and runs the tape / optimizer around it (demo/demo_gcn.py:64-83); on
    small graphs a step is launch-bound there and here alike (a 2-layer GCN step at ogbn-arxiv shape is ≈ 40 launches of
    5-150 us each, issued by Python + autograd).  Every launch of the step sits on the current HIP stream with
    allocator-owned buffers, so the step is a fixed launch sequence and the replay issues it in one call.

        opt = torch.optim.Adam(params, lr=1e-2, capturable=True)        # the step counter must live on the device
        step = tfg.CapturedTrainStep(lambda: loss_of(model([x, ei, w], training=True, cache=cache)), opt)
        for _ in range(epochs):
            loss = step()               # a device scalar; reading it (float(loss)) is the only synchronisation

    loss_fn() must read its data from tensors that keep their addresses (x, labels, index tensors), over plans that are
    already cached (run the model once eagerly first).  The warm-up steps this constructor needs (they size the allocator
    pools, build lazily-built plan metadata and create the optimizer state) are rolled back: parameters and optimizer
    state are what they were before the constructor ran, so step k of the replay equals step k of the eager loop.

    Randomness: torch's dropout / rand_like draw from a device-side Philox offset that a replay advances, and the attention
    dropout of GAT layers reads its seed from a device tensor the captured sequence advances (nn/conv/gat.new_drop_seed) —
    every replay draws fresh masks.  (A by-value seed would be frozen into the graph; tests/test_gpu_regressions.py holds
    the replays to differ and to match the eager layer run with each replay's seed.)
    """

    def __init__(self, loss_fn, optimizer, warmup=3):
        L.require_gpu()
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"]:
                raise ValueError("CapturedTrainStep: construct the optimizer with capturable=True (its step counter has "
                                 "to be a device tensor to be advanced by a replay)")
        self.optimizer = optimizer
        params = [p for g in optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        saved_s = {id(p): {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                   for p, st in optimizer.state.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        from .plan import no_auto_promotion
        with torch.cuda.stream(side), no_auto_promotion():      # a layout promoted here could never be replayed (static_rows)
            for _ in range(max(int(warmup), 1)):
                optimizer.zero_grad(set_to_none=True)
                loss_fn().backward()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():                       # roll the warm-up back, in place (addresses are what gets captured)
            for p, s in zip(params, saved_p):
                p.copy_(s)
            for p, st in optimizer.state.items():
                before = saved_s.get(id(p), {})
                for k, v in st.items():
                    if isinstance(v, torch.Tensor):
                        if isinstance(before.get(k), torch.Tensor):
                            v.copy_(before[k])
                        else:
                            v.zero_()               # state the warm-up created: Adam's moments / step, SGD's momentum
        optimizer.zero_grad(set_to_none=True)       # gradients are (re)allocated inside the capture, from the graph's pool
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = loss_fn()
            self.loss.backward()
            optimizer.step()

    def __call__(self):
        self.graph.replay()
        return self.loss
