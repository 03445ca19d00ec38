# coding=utf-8
"""torch.segment_reduce (sorted segments, no atomics) vs index_add / scatter_reduce in float64: value, gradient, time —
uniform and R-MAT destination distributions at Reddit size."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from tf_geometric_amd import synthetic, _lib as L

n, e, _ = synthetic.WORKLOADS["reddit"]
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, time.perf_counter() - t
for name in ("uniform", "rmat"):
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=3)) if name == "uniform" else synthetic.rmat_edges(n, e, 13, torch.device("cuda"))
    row = ei[0].long()
    order = torch.sort(row, stable=True).indices
    lengths = torch.bincount(row, minlength=n)
    print(name, "max in-degree", int(lengths.max()))
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for c in (1, 8):
        data = torch.randn(int(row.shape[0]), c, generator=g, device="cuda", dtype=torch.float64).requires_grad_()
        up = torch.randn(n, c, generator=g, device="cuda", dtype=torch.float64)
        (s1, t1) = T(lambda: torch.segment_reduce(data[order], "sum", lengths=lengths, axis=0, unsafe=True))
        _, t1b = T(lambda: s1.backward(up))
        g1 = data.grad.clone(); data.grad = None
        if name == "uniform" or c == 1:
            (s2, t2) = T(lambda: torch.zeros(n, c, dtype=torch.float64, device="cuda").index_add(0, row, data))
            _, t2b = T(lambda: s2.backward(up))
            print("  sum c={}: segment_reduce fwd {:.3f}s bwd {:.3f}s | index_add fwd {:.3f}s bwd {:.3f}s | value diff {:.2e} grad diff {:.2e}".format(
                c, t1, t1b, t2, t2b, float((s1 - s2).abs().max()), float((g1 - data.grad).abs().max())))
            data.grad = None
        else:
            print("  sum c={}: segment_reduce fwd {:.3f}s bwd {:.3f}s".format(c, t1, t1b))
        (m1, t3) = T(lambda: torch.segment_reduce(data[order], "max", lengths=lengths, axis=0, unsafe=True, initial=-3.4e38))
        _, t3b = T(lambda: m1.backward(up))
        print("  max c={}: segment_reduce fwd {:.3f}s bwd {:.3f}s".format(c, t3, t3b))
        data.grad = None
        del data, up
