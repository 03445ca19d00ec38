# coding=utf-8
"""End-time spread of the fused aggregate -> GEMM launch's workgroups (library built with -DTFGX_FUSED_DEBUG, see
tools/build_variant.sh): every wave stores its entry / exit tick of the LAST launch.

    TFGX_LIB_PATH=tf_geometric_amd/lib/variants/fdbg/libtfgx.so python tools/fused_wave_dump.py [products|arxiv] [uniform|rmat]
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                          # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P     # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "products"
graph = sys.argv[2] if len(sys.argv) > 2 else "uniform"
n, e, f = synthetic.WORKLOADS[which]
ei = synthetic.rmat_edges(n, e, 7, torch.device("cuda")) if graph == "rmat" else L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
cache = {}
P.AUTO_STATIC_LAYOUT = False
gcn = tfg.layers.GCN(256, activation=tfg.relu)
lib = L.require_gpu()
fn = lib.tfgx_debug_fused_waves
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(4):
    gcn([x, ei], cache=cache)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
gcn([x, ei], cache=cache)
e1.record()
torch.cuda.synchronize()
nw = 256 * 16
arr = np.zeros((nw, 2), dtype=np.uint64)
fn(arr.ctypes.data, nw)
ok = arr[:, 1] > 0
t0 = arr[ok, 0].min()
end = (arr[ok, 1] - t0) / 100.0
wg_end = np.array([end[i * 16:(i + 1) * 16].max() for i in range(int(ok.sum()) // 16)])
print(json.dumps({"workload": which, "graph": graph, "layer_ms": e0.elapsed_time(e1), "waves": int(ok.sum()),
                  "wg_end_us_min_p10_p50_p90_max": [float(np.percentile(wg_end, q)) for q in (0, 10, 50, 90, 100)],
                  "wg_end_by_xcd_mean": [round(float(wg_end[i::8].mean()), 1) for i in range(8)],
                  "entry_skew_us": float((arr[ok, 0].max() - t0) / 100.0)}))
