# coding=utf-8
"""Import-time placeholder for the reference's optional ``ogb_lite`` dependency (tf_geometric/datasets/ogb.py:6)."""
