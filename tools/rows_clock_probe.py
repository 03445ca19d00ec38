# coding=utf-8
"""In-kernel clocks of the row-streaming GEMM (a library built with -DTFGX_ROWS_EXPERIMENT=3, see tools/build_variant.sh):
shader-clock cycles and 100 MHz ticks each wave spends in its tile loop -> the clock the part really runs at under this
kernel, and cycles per tile against the MFMA-only figure (K/2 * TN * 64).

    TFGX_LIB_PATH=tf_geometric_amd/lib/variants/dbg3/libtfgx.so python tools/rows_clock_probe.py
"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import _lib as L                  # noqa: E402
from tf_geometric_amd.plan import gemm_bias_act         # noqa: E402

lib = L.require_gpu()
fn = lib.tfgx_debug_rows_stats
fn.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
fn.restype = ctypes.c_int
buf = (ctypes.c_uint64 * 16)()
g = torch.Generator(device="cuda")
g.manual_seed(0)
for M, K, N in [(2400000, 256, 128), (2400000, 100, 128), (2400000, 128, 256), (2400000, 100, 256), (2400000, 256, 40)]:
    a = torch.randn(M, K, generator=g, device="cuda")
    b = torch.randn(K, N, generator=g, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        gemm_bias_act(a, b, out=out)
    fn(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 1
    e0.record()
    for _ in range(reps):
        gemm_bias_act(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    fn(buf)
    cyc, ticks, tiles, waves, tmax, tmin, tpro, tspan, k0, k1, e0_, e1_, cmf, cep, cw0, cw1 = [int(v) for v in buf][:16]
    tn = (N + 31) // 32
    groups = (K // 32) * 16 + (K % 32) // 2
    ideal = groups * tn * 64
    print(json.dumps({"M": M, "K": K, "N": N, "ms": e0.elapsed_time(e1) / reps, "waves": waves // reps, "tiles_per_wave": tiles / waves,
                      "shader_clock_GHz": cyc / ticks * 0.1, "cycles_per_tile_per_wave": cyc / tiles,
                      "mfma_cycles_per_tile": ideal, "two_waves_share_a_pipe": 2 * ideal,
                      "pipe_busy_est": 2 * ideal / (cyc / tiles),
                      "loop_us_avg_min_max": [ticks / waves / 100.0, tmin / 100.0, tmax / 100.0], "prologue_us_avg": tpro / waves / 100.0,
                      "wave_start_to_end_us_max": tspan / 100.0,
                      "kernel_entry_skew_us": (k1 - k0) / 100.0, "loop_end_skew_us": (e1_ - e0_) / 100.0,
                      "first_entry_to_last_end_us": (e1_ - k0) / 100.0,
                      "cycles_per_tile_in_full_step_mfma_groups": cmf / tiles, "cycles_per_tile_in_epilogue": cep / tiles,
                      "cycles_per_tile_elsewhere": (cyc - cmf - cep) / tiles,
                      "cycles_per_tile_handover_after_first_step": cw0 / tiles, "cycles_per_tile_handover_other_steps": cw1 / tiles,
                      "cycles_per_mfma_inside_groups_per_wave": cmf / tiles / ((K // 32) * 16 * tn)}), flush=True)
