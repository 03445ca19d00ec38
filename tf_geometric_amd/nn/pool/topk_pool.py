# coding=utf-8
"""topk_pool.  Reference: tf_geometric/nn/pool/topk_pool.py:6-87 — a dense
[num_sources, max_targets] score matrix, a row-wise argsort and a mask; here one tfgx_segment_topk call
(64-bit radix sort + scan + masked copy, O(n) memory)."""
import torch

from ... import _lib as L


def topk_pool(source_index, score, k=None, ratio=None):
    """
    :param source_index: index of source node (of edge) or source graph (of node)
    :param score: 1-D Array
    :param k: Keep top k targets for each source
    :param ratio: Keep num_targets * ratio targets for each source
    :return: positions (into source_index / score) of the kept targets: sources ascending, scores descending,
        ties in input order.  ratio > 1 keeps every target (the reference would index its padding columns there).
    """
    if k is None and ratio is None:
        raise Exception("you should provide either k or ratio for topk_pool")
    elif k is not None and ratio is not None:
        raise Exception("you should provide either k or ratio for topk_pool, not both of them")
    lib = L.require_gpu()
    as_np = not isinstance(source_index, torch.Tensor)
    seg = L.as_i32(source_index).reshape(-1)
    sc = L.as_f32(score).detach().reshape(-1)
    n = int(seg.shape[0])
    if int(sc.shape[0]) != n:
        raise Exception("topk_pool: source_index and score differ in length")
    dev = seg.device
    if n == 0:
        out = torch.zeros(0, dtype=torch.int32, device=dev)
        return out.cpu().numpy() if as_np else out
    num_segments = int(seg.max().item()) + 1
    out_index = torch.empty(n, dtype=torch.int32, device=dev)
    out_count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.tfgx_segment_topk_workspace_bytes(n, num_segments)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.tfgx_segment_topk(L.ptr(seg), L.ptr(sc), n, num_segments, -1 if k is None else int(k),
                                  0.0 if ratio is None else float(ratio), L.ptr(out_index), L.ptr(out_count), L.ptr(ws),
                                  ws_bytes, L.stream_ptr()), "tfgx_segment_topk")
    out = out_index[:int(out_count.item())]
    return out.cpu().numpy() if as_np else out
