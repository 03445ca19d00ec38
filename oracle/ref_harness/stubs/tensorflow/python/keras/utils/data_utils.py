# coding=utf-8
"""Import-time placeholder (tf_geometric/data/dataset.py:4); dataset download/extraction is out of scope."""


def _extract_archive(file_path, path=".", archive_format="auto"):
    raise NotImplementedError("dataset archives are not handled by the numpy stand-in for tensorflow")
