# coding=utf-8
"""A few sharded aggregation steps in self-halo mode (one GPU: 7/8 of the source rows travel through the product transport —
pack kernel, grouped ncclSend / ncclRecv on the communication stream — in 4 rounds while the passes run on the compute
stream) for `rocprofv3 --kernel-trace`: tools/trace_overlap.py then measures how much of the exchange kernels' time overlaps
the reduce passes."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29743")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from tf_geometric_amd import synthetic, _lib as L                     # noqa: E402
from tf_geometric_amd.dist.sharded import ShardedGraph                # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
W, R = 8, 4
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
sg = ShardedGraph.from_global(ei, n, rounds=R, self_halo_rows=n // W)
table = sg.alloc_table(f)
sg.own_rows(table).copy_(torch.randn(n, f, device="cuda"))
out = torch.empty(sg.n_own, f, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "forward"              # "train": forward + backward (reverse exchange) per step
x_own = torch.randn(n, f, device="cuda", requires_grad=True)
g_out = torch.randn(sg.n_own, f, device="cuda")


def step():
    if mode == "train":
        x_own.grad = None
        sg.aggregate_trainable(x_own, L.SUM, w=None).backward(g_out)
    else:
        sg.aggregate(table, L.SUM, w=None, out=out)


for _ in range(3):
    step()
torch.cuda.synchronize()
import time                                                           # noqa: E402
time.sleep(0.3)                                                       # an idle gap: trace_overlap.py cuts phases at it
for _ in range(5):
    step()
torch.cuda.synchronize()
from tf_geometric_amd.dist.transport import close_transports         # noqa: E402
close_transports()
dist.destroy_process_group()
print("done")
