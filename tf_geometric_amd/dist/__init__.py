# coding=utf-8
"""Sharded path: dst-range shards (sharded.py) + the transports that move halo rows (transport.py).

Multi-process GPU work on this platform needs dmabuf IPC: the host driver of the MI355X boxes does not support the legacy
IPC mode, and without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL / device-memory sharing across processes fails inside
hipIpcGetMemHandle ("invalid argument").  The ROCr runtime reads the variable when it initialises (the first HIP call of
the process), so it is defaulted HERE, at import time of the package that owns every multi-rank code path — `setdefault`:
a value the user exported wins.  transport.ipc_mode_note() reports what the process actually runs with; bench.py sets the
same default before it imports torch and passes it on to the ranks it launches.
"""
import os as _os

_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
