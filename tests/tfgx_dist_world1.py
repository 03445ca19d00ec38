# coding=utf-8
"""The PRODUCT transport of the sharded path (dist/transport.py::TfgxDistTransport -> lib/libtfgx_dist.so -> RCCL) under a
one-rank "nccl" torch.distributed group on one GPU, in self-halo test mode: only a third of the rank's rows are resident
sources, the rest are requested from the rank itself, so packed rows really travel through grouped ncclSend / ncclRecv on
the communication stream, round by round, forward and reverse.  Run by tests/test_gpu_dist.py in a subprocess (two ranks
cannot share a GPU under RCCL; the multi-rank exchange first runs on the driver's 8-GPU node)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29735")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")

import dist_worker                                   # noqa: E402
from conftest import assert_parity                   # noqa: E402

for partitioned in (False, True):
    res = {}
    dist_worker.run_checks(0, 1, use_gpu=True, skew=True, results=res, rounds=3, partitioned=partitioned, self_halo=True)
    p = res[0]
    assert p["transport"] == "tfgx_dist" and p["rows_sent"] > 0 and p["rows_packed"] == p["rows_sent"] and p["n_halo"] > 0, p
    dist_worker.check_against_reference([p], True, assert_parity)
    print("forward layers through tfgx_dist ok (partitioned = {}): {} halo rows, {} rows sent per exchange".format(
        partitioned, p["n_halo"], p["rows_sent"]))
tr = dist_worker.run_training(0, 1, True, False, rounds=3, num_splits=4, self_halo=True)
ref = dist_worker.training_reference(False)
assert_parity(tr["out"], ref["out"], what="trainable forward")
assert_parity(tr["dx"], ref["dx"], tol=2e-5, what="d/dx through the reverse exchange")
assert_parity(tr["dx_mean"], ref["dx_mean"], tol=2e-5, what="mean d/dx")
assert_parity(tr["dk"], ref["dk"], tol=1e-4, what="d/dkernel")
dist_worker.check_training_extras([tr], ref, assert_parity)
assert np.array_equal(tr["chunked"], tr["whole"])            # column-chunked halo: two exchanges in flight on two plans
# the trainable max / max-pool SAGE / GAT forwards ran span by span under the exchange (not exchange-then-one-pass)
assert tr["counters"].get("max_span_forwards", 0) >= 2 and tr["counters"].get("gat_span_training_forwards", 0) >= 1, tr["counters"]
print("training through tfgx_dist ok")
torch.cuda.synchronize()
from tf_geometric_amd.dist import transport as T               # noqa: E402
from tf_geometric_amd.dist.transport import close_transports   # noqa: E402

# the agreed safety net of auto mode (world > 1 takes it): the communicator's self-check, and the fallback to torch's own
# RCCL collectives when the C-ABI transport cannot be brought up on some rank
from tf_geometric_amd.dist.sharded import HipBackend           # noqa: E402
chk = T._bring_up_tfgx(None, HipBackend(), auto=False)
assert chk.name == "tfgx_dist" and chk.comm_info()[:2] == (1, 0), chk.comm_info()       # ncclCommCount / UserRank
chk.close()


class _Broken(T.TfgxDistTransport):
    def self_check(self):
        raise RuntimeError("injected")


real, T.TfgxDistTransport = T.TfgxDistTransport, _Broken
import warnings                                                # noqa: E402
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    fb = T._bring_up_tfgx(None, HipBackend(), auto=True)
    try:                                  # asked for by name (bench.py, TFGX_DIST_TRANSPORT): no net, it raises
        T._bring_up_tfgx(None, HipBackend(), auto=False)
        raise AssertionError("strict mode fell back")
    except T.TfgxDistUnavailable as ex:
        assert "injected" in str(ex)
T.TfgxDistTransport = real
assert fb.name.startswith("torch (fallback") and "injected" in fb.fallback_reason and caught, (fb.name, caught)
got = fb.all_to_all_v(torch.arange(6, dtype=torch.int32, device="cuda").view(3, 2), [3], [3])
assert got.tolist() == [[0, 1], [2, 3], [4, 5]]
print("self-check and agreed fallback ok")
close_transports()
dist.barrier()
dist.destroy_process_group()
print("TFGX_DIST_WORLD1_OK")
