# coding=utf-8
"""How far is the HIP path from the reference's golden outputs, per output, in units of the plain band
1e-5 + 1e-5*|ref|?  (Evidence for every widened tolerance in tests/reference_cases.py: a ratio <= 1 means the plain
band holds and no widening is needed.)    python tools/golden_margin.py > gpurun_out/r03/golden_margin.jsonl"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import reference_cases as rc        # noqa: E402
import tf_geometric_amd as tfg      # noqa: E402

golden = dict(np.load(os.path.join(ROOT, "tests", "golden", "reference_cases.npz")))
for case in rc.CASES:
    if case.hip is None:
        continue
    g = case.inputs()
    got = case.hip(tfg, g, golden) if case.name == "layers" else case.hip(tfg, g)
    for full in [k for k in golden if k.startswith(case.name + "::")]:
        k = full[len(case.name) + 2:]
        a, b = np.asarray(got[k]), golden[full]
        if k in case.exact or a.dtype.kind in "iub" or b.dtype.kind not in "f":
            continue
        a, b = a.astype(np.float64), b.astype(np.float64)
        ratio = float((np.abs(a - b) / (1e-5 + 1e-5 * np.abs(b))).max()) if a.size else 0.0
        if ratio > 0.3 or case.tol_of(k) != rc.TOL:
            print(json.dumps({"case": case.name, "output": k, "tol_in_test": case.tol_of(k),
                              "max_err_over_plain_band": round(ratio, 4), "max_abs_ref": float(np.abs(b).max())}), flush=True)
