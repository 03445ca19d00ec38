# coding=utf-8
"""RCCL with one rank on one GPU: the exact torch.distributed calls dist/sharded.py and bench.py make on the "nccl"
backend — all_to_all_single with split lists into a VIEW of the source table (async + wait), int64 plan-time
exchanges, empty rounds, all_reduce (SUM / MAX), barrier.  Two ranks cannot share a GPU under RCCL, so this is as much
of the RCCL transport as a 1-GPU box can exercise; run by tests/test_gpu_dist.py in a subprocess."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
t = time.time(); dist.init_process_group("nccl"); print("init", round(time.time() - t, 2), dist.get_backend())
x = torch.arange(12, dtype=torch.float32, device="cuda").reshape(4, 3)
table = torch.zeros(10, 3, device="cuda")
halo = table[6:10]
w = dist.all_to_all_single(halo, x, [4], [4], async_op=True); w.wait(); torch.cuda.synchronize()
print("a2a f32 view ok", bool(torch.equal(table[6:], x)))
a = torch.arange(5, dtype=torch.int64, device="cuda"); b = torch.empty(5, dtype=torch.int64, device="cuda")
dist.all_to_all_single(b, a, [5], [5]); print("a2a i64 ok", bool(torch.equal(a, b)))
e0 = torch.empty(0, 3, device="cuda"); e1 = torch.empty(0, 3, device="cuda")
w = dist.all_to_all_single(e1, e0, [0], [0], async_op=True); w.wait(); print("a2a empty ok")
d = torch.ones(3, dtype=torch.int64, device="cuda"); dist.all_reduce(d); print("allreduce", d.tolist())
tt = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); print("max", tt.item())
dist.barrier(); print("barrier ok"); dist.destroy_process_group(); print("RCCL_WORLD1_OK")
