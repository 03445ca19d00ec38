# coding=utf-8
"""Writes tests/golden/reference_cases.npz: the outputs of the reference's OWN Python on the inputs of
tests/reference_cases.py.

The reference (``/root/reference/tf_geometric``, v0.1.7) is imported UNMODIFIED by oracle/ref_harness.  With real
``tensorflow`` + ``tf_sparse`` installed they are used; in this image they are not installable, so the reference
runs on the numpy stand-ins of oracle/ref_harness/stubs (restated TF / tf_sparse primitives; the composition logic —
every line of tf_geometric — is the reference's).  The backend used is recorded in the file (``__backend__``).

    python tests/golden/make_golden_from_reference.py          # regenerate (needs /root/reference)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.ref_harness import load_reference   # noqa: E402
import reference_cases as rc                     # noqa: E402

OUT = os.path.join(HERE, "reference_cases.npz")


def run_reference():
    tfg, tf, tfs, backend = load_reference()
    R = types.SimpleNamespace(tfg=tfg, tf=tf, tfs=tfs)
    blob = {"__backend__": np.array(backend), "__reference_version__": np.array("tf_geometric 0.1.7")}
    for case in rc.CASES:
        outs = case.ref(R, case.inputs())
        for k, v in outs.items():
            blob["{}::{}".format(case.name, k)] = np.asarray(v)
        print("{:<28s} {:3d} outputs".format(case.name, len(outs)))
    return blob


if __name__ == "__main__":
    blob = run_reference()
    np.savez_compressed(OUT, **blob)
    print("wrote {} ({} arrays, {:.0f} KB, backend = {})".format(OUT, len(blob), os.path.getsize(OUT) / 1024.0,
                                                                  blob["__backend__"]))
