# coding=utf-8
"""Per-wave loop start / end / tile count of ONE launch of the row kernel (library built with -DTFGX_ROWS_EXPERIMENT=3)."""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import _lib as L                  # noqa: E402
from tf_geometric_amd.plan import gemm_bias_act         # noqa: E402
lib = L.require_gpu()
fn = lib.tfgx_debug_rows_waves
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, K, N = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2400000,128,256").split(",")]
a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda") * 0.1; out = torch.empty(M, N, device="cuda")
for _ in range(5):
    gemm_bias_act(a, b, out=out)
arr = np.zeros((2048, 4), dtype=np.uint64)
fn(arr.ctypes.data, 2048)
t0 = arr[:, 0].min()
st, en, tiles, xcc = (arr[:, 0] - t0) / 100.0, (arr[:, 1] - t0) / 100.0, arr[:, 2].astype(int), arr[:, 3].astype(int)
wave = np.arange(2048) % 8
wg = np.arange(2048) // 8
print(json.dumps({"shape": [M, K, N], "dynamic": os.environ.get("TFGX_ROWS_DYNAMIC", "0"),
                  "end_us_by_wave_slot": [round(float(en[wave == k].mean()), 1) for k in range(8)],
                  "tiles_by_wave_slot": [round(float(tiles[wave == k].mean()), 2) for k in range(8)],
                  "end_us_by_xcc": {int(x): round(float(en[xcc == x].mean()), 1) for x in np.unique(xcc)},
                  "tiles_by_xcc": {int(x): round(float(tiles[xcc == x].mean()), 2) for x in np.unique(xcc)},
                  "wgs_by_xcc": {int(x): int((xcc[::8] == x).sum()) for x in np.unique(xcc)},
                  "end_us_min_max": [float(en.min()), float(en.max())], "start_us_max": float(st.max()),
                  "us_per_tile_by_wave_slot": [round(float(((en - st) / np.maximum(tiles, 1))[wave == k].mean()), 2) for k in range(8)],
                  "us_per_tile_by_xcc": {int(x): round(float(((en - st) / np.maximum(tiles, 1))[xcc == x].mean()), 2) for x in np.unique(xcc)},
                  "end_us_percentiles_1_50_99": [float(np.percentile(en, q)) for q in (1, 50, 99)]}))
