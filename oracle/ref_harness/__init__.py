# coding=utf-8
"""Harness that imports the reference's own Python (unmodified, from /root/reference) — test infrastructure only."""
from .load_reference import load_reference, reference_available   # noqa: F401
