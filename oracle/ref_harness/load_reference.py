# coding=utf-8
"""
Imports ``tf_geometric`` UNMODIFIED from ``/root/reference`` (or ``$TFG_REFERENCE_ROOT``).

TEST INFRASTRUCTURE ONLY: used by ``tests/golden/make_golden_from_reference.py`` (which writes the committed golden
vectors) and by the non-GPU test ``tests/test_oracle_vs_reference.py``.  Nothing under ``tf_geometric_amd/``, no
``-m gpu`` test, ``bench.py`` or ``smoke()`` may call it: the reference checkout does not exist on the GPU box.

If real ``tensorflow`` and ``tf_sparse`` are importable they are used.  In this image they are not (no wheel, no
network), so the numpy stand-ins under ``stubs/`` are put on ``sys.path`` first: the reference's composition logic
then runs as written, on restated TF / tf_sparse primitives (see the headers of ``stubs/tensorflow/__init__.py`` and
``stubs/tf_sparse/__init__.py`` for exactly which semantics are restated).
"""
import importlib
import importlib.util
import os
import sys
import warnings

_HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(_HERE, "stubs")
REFERENCE_ROOT = os.environ.get("TFG_REFERENCE_ROOT", "/root/reference")

_loaded = {}


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "tf_geometric"))


def _real_tensorflow_present():
    try:
        spec_tf = importlib.util.find_spec("tensorflow")
        spec_tfs = importlib.util.find_spec("tf_sparse")
    except (ImportError, ValueError):
        return False
    return spec_tf is not None and spec_tfs is not None


def load_reference():
    """Returns ``(tfg, tf, tfs, backend)``: the reference package, the tensorflow / tf_sparse modules it runs on, and
    ``backend`` = "tensorflow" (real) or "numpy-stub"."""
    if _loaded:
        return _loaded["tfg"], _loaded["tf"], _loaded["tfs"], _loaded["backend"]
    if not reference_available():
        raise RuntimeError("reference checkout not found at {}".format(REFERENCE_ROOT))
    backend = "tensorflow" if _real_tensorflow_present() else "numpy-stub"
    if backend == "numpy-stub":
        sys.path.insert(0, STUBS)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tf = importlib.import_module("tensorflow")
            tfs = importlib.import_module("tf_sparse")
            tfg = importlib.import_module("tf_geometric")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        if backend == "numpy-stub":
            sys.path.remove(STUBS)
    assert os.path.abspath(tfg.__file__).startswith(os.path.abspath(REFERENCE_ROOT)), tfg.__file__
    _loaded.update(tfg=tfg, tf=tf, tfs=tfs, backend=backend)
    return tfg, tf, tfs, backend
