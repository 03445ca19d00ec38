# coding=utf-8
"""
CPU oracle for the tf_geometric message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``tf_geometric_amd/`` may import this
module: it is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``, never the thing that is shipped or timed
as the product.

PINNED TO THE REFERENCE'S OWN PYTHON (round 2).  The reference (/root/reference, tf_geometric 0.1.7) ships no
tests, golden vectors or fixtures for this path and hard-depends on two third-party packages that
are not installable here: ``tensorflow`` (>=1.15,<2 or >=2.4.0, setup.py:33-38) and ``tf_sparse >= 0.0.17``
(setup.py:25).  ``oracle/ref_harness`` therefore provides numpy stand-ins for exactly those two modules and imports
``/root/reference/tf_geometric`` UNMODIFIED on top of them: every line of the reference's composition logic
(aggregate_neighbors, segment_softmax, gcn_norm_adj, gcn, gat, the GraphSAGE variants, graph_utils, the tfg.layers
classes, the other convs and pools) executes as written and produces ``tests/golden/reference_cases.npz``
(``tests/golden/make_golden_from_reference.py``).  ``tests/test_oracle_vs_reference.py`` holds this file to those
outputs (and, where /root/reference exists, to the live reference on further fuzz seeds).  That check already found one
deviation of the first-round restatement: ``aggregate_neighbors`` with a ``[2, 0]`` edge_index does NOT take the
early return of map_reduce.py:57.

What is still RESTATED rather than executed — in the stand-ins, and equally here — is the arithmetic of the TF /
tf_sparse primitives at the call sites (TensorFlow itself has never run in this image):

  * tf.gather(params, idx)                      -> params[idx]
  * tf.math.unsorted_segment_sum(d, ids, n)     -> zeros(n) then out[ids[i]] += d[i]; empty segment = 0
  * tf.math.unsorted_segment_mean(d, ids, n)    -> segment_sum / max(count, 1); empty segment = 0
  * tf.math.unsorted_segment_max(d, ids, n)     -> empty segment = numeric_limits<float>::lowest()
  * tf.nn.l2_normalize(x, axis, eps=1e-12)      -> x * rsqrt(max(sum(x**2), eps))
  * tf.unique                                   -> first-occurrence order
  * tf_sparse.SparseMatrix(index, value, shape) -> COO matrix, duplicate entries are summed by matmul
  * SparseMatrix.segment_sum(axis=-1 | 0)       -> row sums | column sums of the values
  * SparseMatrix.add_diag(c)                    -> A + c*I, existing diagonal entries added to (restated here as N
                                                   appended (i,i,c) entries: identical under matmul / segment_sum)
  * SparseMatrix.segment_softmax(axis=-1)       -> nn/kernel/segment.py:26-33 grouped by row
  * SparseMatrix.dropout(rate, training=False)  -> identity

Further pins (tests/test_oracle.py): hand-derived known answers on the reference's own example graphs
(tutorial_intro.py:24-30, datasets/synthetic.py:47-66), an independent second implementation (torch CPU index_add_ /
scatter_reduce and the C restatement oracle/tfg_oracle.c), and algebraic properties (edge-order permutation
invariance, duplicate edges sum, linearity).

Two accumulation modes: ``acc=np.float64`` (default; the "true" value the fp32
implementations are compared with under 1e-5 + 1e-5*|ref|) and
``acc=np.float32`` (op-for-op fp32, what TF-CPU computes).
"""
import numpy as np

FLT_LOWEST = np.float32(-3.4028234663852886e38)  # std::numeric_limits<float>::lowest()


# ----------------------------------------------------------------------------
# segment ops  (TF semantics restated; callers: nn/kernel/map_reduce.py:15-42)
# ----------------------------------------------------------------------------

def _as2d(data):
    data = np.asarray(data)
    if data.ndim == 1:
        return data[:, None], True
    return data, False


def unsorted_segment_sum(data, segment_ids, num_segments, acc=np.float64):
    """tf.math.unsorted_segment_sum as used at nn/kernel/map_reduce.py:16."""
    data, squeeze = _as2d(data)
    ids = np.asarray(segment_ids).astype(np.int64)
    if ids.size and (ids.min() < 0 or ids.max() >= num_segments):
        raise ValueError("segment id out of range")  # TF-CPU raises InvalidArgumentError
    out = np.zeros((num_segments, data.shape[1]), dtype=acc)
    if ids.size:
        order = np.argsort(ids, kind="stable")
        sid = ids[order]
        starts = np.flatnonzero(np.concatenate(([True], sid[1:] != sid[:-1])))
        red = np.add.reduceat(data[order].astype(acc), starts, axis=0)
        out[sid[starts]] = red
    out = out.astype(np.float32) if data.dtype == np.float32 else out
    return out[:, 0] if squeeze else out


def segment_count(index, num_segments=None):
    """nn/kernel/segment.py:36-40."""
    index = np.asarray(index)
    if num_segments is None:
        num_segments = int(index.max()) + 1
    return np.bincount(index.astype(np.int64), minlength=num_segments).astype(index.dtype)


def unsorted_segment_mean(data, segment_ids, num_segments, acc=np.float64):
    """tf.math.unsorted_segment_mean as used at nn/kernel/map_reduce.py:28: sum / max(count, 1)."""
    data2, squeeze = _as2d(data)
    s = unsorted_segment_sum(np.asarray(data2, dtype=acc), segment_ids, num_segments, acc=acc)
    cnt = np.bincount(np.asarray(segment_ids).astype(np.int64), minlength=num_segments)
    out = (s / np.maximum(cnt, 1)[:, None].astype(acc)).astype(np.float32)
    return out[:, 0] if squeeze else out


def unsorted_segment_max(data, segment_ids, num_segments):
    """tf.math.unsorted_segment_max as used at nn/kernel/map_reduce.py:41 and
    nn/kernel/segment.py:27; an empty segment holds float32 lowest()."""
    data, squeeze = _as2d(data)
    ids = np.asarray(segment_ids).astype(np.int64)
    out = np.full((num_segments, data.shape[1]), FLT_LOWEST, dtype=np.float32)
    if ids.size:
        order = np.argsort(ids, kind="stable")
        sid = ids[order]
        starts = np.flatnonzero(np.concatenate(([True], sid[1:] != sid[:-1])))
        red = np.maximum.reduceat(data[order].astype(np.float32), starts, axis=0)
        out[sid[starts]] = red
    return out[:, 0] if squeeze else out


def sorted_segment(kind, data, segment_ids, acc=np.float64):
    """tf.math.segment_sum / segment_mean / segment_max / segment_min (the callables nn/kernel/segment.py:5 is handed at
    nn/kernel/map_reduce.py:35 and nn/pool/common_pool.py:28,36): ids ascending, output rows = last id + 1, a segment id
    that does not occur yields 0 (TensorFlow's documented value — unlike the unsorted max / min, which hold lowest / max)."""
    data2, squeeze = _as2d(data)
    ids = np.asarray(segment_ids).astype(np.int64)
    if ids.size and (np.diff(ids) < 0).any():
        raise ValueError("segment ids are not increasing")
    n = int(ids[-1]) + 1 if ids.size else 0
    out = np.zeros((n, data2.shape[1]), dtype=np.float32)
    if ids.size:
        starts = np.flatnonzero(np.concatenate(([True], ids[1:] != ids[:-1])))
        if kind in ("sum", "mean"):
            red = np.add.reduceat(data2.astype(acc), starts, axis=0)
            if kind == "mean":
                red = red / np.diff(np.concatenate((starts, [ids.size])))[:, None]
        else:
            red = (np.maximum if kind == "max" else np.minimum).reduceat(data2.astype(np.float32), starts, axis=0)
        out[ids[starts]] = red.astype(np.float32)
    return out[:, 0] if squeeze else out


def segment_op_with_pad(segment_op, data, segment_ids, num_segments):
    """nn/kernel/segment.py:5-23 line by line; segment_op(sorted_data, sorted_ids) is a sorted segment op, e.g.
    functools.partial(sorted_segment, "max")."""
    data = np.asarray(data)
    segment_ids = np.asarray(segment_ids)
    sort_index = np.argsort(segment_ids, kind="stable")                    # :7
    sorted_segment_ids = segment_ids[sort_index]                           # :8
    sorted_data = data[sort_index]                                         # :9
    reduced_data = segment_op(sorted_data, sorted_segment_ids)             # :11
    num_paddings = num_segments - reduced_data.shape[0]                    # :12
    pads = np.zeros((num_paddings,) + data.shape[1:], dtype=reduced_data.dtype)    # :14-18
    return np.concatenate([reduced_data, pads], axis=0)                    # :19-23


def segment_softmax(data, segment_ids, num_segments, acc=np.float64):
    """nn/kernel/segment.py:26-33 line by line."""
    data = np.asarray(data, dtype=np.float32)
    ids = np.asarray(segment_ids).astype(np.int64)
    max_values = unsorted_segment_max(data, ids, num_segments)            # :27
    gathered_max = max_values[ids]                                         # :28
    ex = np.exp((data - gathered_max).astype(acc))                         # :29
    denom = unsorted_segment_sum(ex, ids, num_segments, acc=acc) + 1e-8    # :30
    score = ex / denom[ids]                                                # :31-32
    return score.astype(np.float32)


# ----------------------------------------------------------------------------
# mappers / reducers / updaters  (nn/kernel/map_reduce.py:7-42, nn/conv/gcn.py:221-222)
# ----------------------------------------------------------------------------

def identity_mapper(repeated_x, neighbor_x, edge_weight=None):
    return neighbor_x                                                      # map_reduce.py:7-8


def neighbor_count_mapper(repeated_x, neighbor_x, edge_weight=None):
    return np.ones([neighbor_x.shape[0], 1], dtype=np.float32)            # map_reduce.py:11-12


def gcn_mapper(repeated_x, neighbor_x, edge_weight=None):
    return neighbor_x * np.asarray(edge_weight)[:, None]                   # gcn.py:221-222 (None -> error)


def sum_reducer(neighbor_msg, node_index, num_nodes=None, acc=np.float64):
    return unsorted_segment_sum(neighbor_msg, node_index, num_nodes, acc=acc)   # :15-16


def mean_reducer(neighbor_msg, node_index, num_nodes=None, acc=np.float64):
    return unsorted_segment_mean(neighbor_msg, node_index, num_nodes, acc=acc)  # :27-28


def max_reducer(neighbor_msg, node_index, num_nodes=None, acc=None):
    if num_nodes is None:
        num_nodes = int(np.max(node_index)) + 1                            # :38-40
    return unsorted_segment_max(neighbor_msg, node_index, num_nodes)       # :41


def sum_updater(x, reduced_neighbor_msg):
    return x + reduced_neighbor_msg                                        # :19-20


def identity_updater(x, reduced_neighbor_msg):
    return reduced_neighbor_msg                                            # :23-24


def aggregate_neighbors(x, edge_index, edge_weight=None, mapper=identity_mapper,
                        reducer=sum_reducer, updater=sum_updater, num_nodes=None, acc=np.float64):
    """nn/kernel/map_reduce.py:45-73 line by line."""
    x = np.asarray(x, dtype=np.float32)
    edge_index = np.asarray(edge_index)
    # :57 `tf.shape(edge_index)[0] == 0` tests the FIRST dimension: true only for "no edges" given as [] / shape [0, ...].
    # A [2, 0] edge_index goes on (checked against the reference itself: empty gathers, then the reducer's value for
    # empty segments — 0 for sum/mean, float32 lowest for max — and the updater)
    if edge_index.shape[0] == 0:
        return x
    edge_index = edge_index.astype(np.int64)
    row, col = edge_index[0], edge_index[1]                                # :60
    repeated_x = x[row]                                                    # :62
    neighbor_x = x[col]                                                    # :63
    msg = mapper(repeated_x, neighbor_x, edge_weight=edge_weight)          # :65
    if num_nodes is None:
        num_nodes = x.shape[0]                                             # :67-68
    reduced = reducer(np.asarray(msg, dtype=np.float32), row, num_nodes=num_nodes, acc=acc)  # :70
    return updater(x, reduced).astype(np.float32)                          # :71


# ----------------------------------------------------------------------------
# tf_sparse.SparseMatrix surface used by the path (restated, see header)
# ----------------------------------------------------------------------------

def add_self_loop_edge(edge_index, num_nodes, edge_weight=None, fill_weight=1.0):
    """utils/graph_utils.py:350-366: APPEND N diagonal edges after the input edges."""
    diag = np.stack([np.arange(num_nodes, dtype=np.int32)] * 2, axis=0)
    ei = np.concatenate([np.asarray(edge_index, dtype=np.int32).reshape(2, -1), diag], axis=1)
    if edge_weight is not None:
        ew = np.concatenate([np.asarray(edge_weight, dtype=np.float32),
                             np.full([num_nodes], fill_weight, dtype=np.float32)])
    else:
        ew = None
    return ei, ew


def spmm(index, value, shape, h, acc=np.float64):
    """SparseMatrix(index, value, shape) @ h : out[r] = sum_{e: row_e = r} value_e * h[col_e]."""
    row, col = np.asarray(index[0]).astype(np.int64), np.asarray(index[1]).astype(np.int64)
    h = np.asarray(h, dtype=np.float32)
    msg = h[col].astype(acc) * np.asarray(value, dtype=np.float32).astype(acc)[:, None]
    return unsorted_segment_sum(msg, row, shape[0], acc=acc).astype(np.float32)


def _remove_inf_and_nan(x):
    return np.where(np.isinf(x) | np.isnan(x), np.zeros_like(x), x)       # gcn.py:23-29


def gcn_norm_adj(edge_index, edge_weight, num_nodes, norm="both", add_self_loop=True, sym=True,
                 renorm=True, improved=False, acc=np.float64, row_deg=None):
    """nn/conv/gcn.py:32-130 for a square adjacency (non-square forbids add_self_loop/sym, :65-69).
    Returns (index [2,E'], value [E'] float32) of the normalised adjacency, self-loops appended.
    row_deg (test infrastructure for CUT-OUT sub-problems of a big graph, tests/test_gpu_fullsize.py): the row sums the
    reference would compute at :80 on the WHOLE graph (diagonal included where the mode adds it first), given for the
    nodes of the sub-problem — a source node's degree depends on in-edges the cut-out does not hold.  norm="both", sym=True
    only; every other line runs unchanged on the sub-problem's edges."""
    ei = np.asarray(edge_index, dtype=np.int32).reshape(2, -1)
    E = ei.shape[1]
    w = np.ones([E], dtype=np.float32) if edge_weight is None else np.asarray(edge_weight, dtype=np.float32)
    fill = 2.0 if improved else 1.0                                        # :62
    N = num_nodes

    def add_diag(ei_, w_):
        ei2, w2 = add_self_loop_edge(ei_, N, w_, fill_weight=fill)
        return ei2, w2

    def row_sum(ei_, w_):
        if row_deg is not None:
            assert norm == "both" and sym and np.shape(row_deg) == (N,)
            return np.asarray(row_deg, dtype=acc)
        return unsorted_segment_sum(w_.astype(acc), ei_[0], N, acc=acc)

    def col_sum(ei_, w_):
        return unsorted_segment_sum(w_.astype(acc), ei_[1], N, acc=acc)

    with np.errstate(divide="ignore", invalid="ignore"):
        if add_self_loop and norm != "both":
            ei, w = add_diag(ei, w)                                        # :71-72
        if norm == "both":
            if add_self_loop and renorm:
                ei, w = add_diag(ei, w)                                    # :76-77
            row_deg = row_sum(ei, w)                                       # :80
            rdis = _remove_inf_and_nan(np.power(row_deg, -0.5))            # :81-82
            if sym:
                cdis = rdis                                                # :85-86
            else:
                cdis = _remove_inf_and_nan(np.power(col_sum(ei, w), -0.5))  # :88-91
            nw = rdis[ei[0]] * w.astype(acc) * cdis[ei[1]]                 # :94
            if add_self_loop and not renorm:
                ei, nw = add_self_loop_edge(ei, N, nw.astype(np.float32), fill_weight=fill)  # :97-98
        elif norm == "left":
            rinv = _remove_inf_and_nan(np.power(row_sum(ei, w), -1.0))     # :103-105
            nw = rinv[ei[0]] * w.astype(acc)                               # :109
        elif norm == "right":
            # quirk kept: "col_deg" is segment_sum(axis=-1), i.e. ROW sums (:113), applied on columns (:119)
            cinv = _remove_inf_and_nan(np.power(row_sum(ei, w), -1.0))
            nw = w.astype(acc) * cinv[ei[1]]
        else:
            raise Exception("wrong GCN norm type: {}".format(norm))        # :122
    return ei, np.asarray(nw, dtype=np.float32)


def _act(name_or_fn, h):
    if name_or_fn is None:
        return h
    if callable(name_or_fn):
        return name_or_fn(h)
    if name_or_fn == "relu":
        return np.maximum(h, 0)
    raise ValueError(name_or_fn)


def matmul(a, b, acc=np.float64):
    return (np.asarray(a, dtype=np.float32).astype(acc) @ np.asarray(b, dtype=np.float32).astype(acc)).astype(np.float32)


def gcn(x, edge_index, edge_weight, kernel, bias=None, activation=None, norm="both", add_self_loop=True,
        sym=True, renorm=True, improved=False, acc=np.float64, row_deg=None):
    """nn/conv/gcn.py:225-290 (inference: dropout is identity, num_or_size_splits does not change the result).
    row_deg: see gcn_norm_adj (cut-out sub-problems only)."""
    x = np.asarray(x, dtype=np.float32)
    N = x.shape[0]
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, norm, add_self_loop, sym, renorm, improved, acc=acc,
                          row_deg=row_deg)                                 # :260
    h = x if kernel is None else matmul(x, kernel, acc)                   # :266-272
    h = spmm(ei, nw, (N, N), h, acc=acc)                                   # :280
    if bias is not None:
        h = h + np.asarray(bias, dtype=np.float32)                         # :284-285
    return _act(activation, h).astype(np.float32)                          # :287-288


def l2_normalize(h, eps=1e-12):
    """tf.nn.l2_normalize(h, axis=-1)."""
    h = np.asarray(h, dtype=np.float64)
    return (h / np.sqrt(np.maximum((h * h).sum(-1, keepdims=True), eps))).astype(np.float32)


def gat(x, edge_index, query_kernel, query_bias, query_activation, key_kernel, key_bias, key_activation,
        kernel, bias=None, activation=None, num_heads=1, split_value_heads=True, acc=np.float64):
    """nn/conv/gat.py:13-122 line by line (inference)."""
    x = np.asarray(x, dtype=np.float32)
    N = x.shape[0]
    ei, _ = add_self_loop_edge(edge_index, N)                              # :43
    row, col = ei[0], ei[1]                                                # :45
    Q = matmul(x, query_kernel, acc) + np.asarray(query_bias, np.float32)  # :52-53
    Q = _act(query_activation, Q)[row]                                     # :54-56
    K = matmul(x, key_kernel, acc) + np.asarray(key_bias, np.float32)      # :61-62
    K = _act(key_activation, K)[col]                                       # :63-65
    V = matmul(x, kernel, acc)                                             # :70
    H = num_heads
    Q_ = np.concatenate(np.split(Q, H, axis=-1), axis=0)                   # :73
    K_ = np.concatenate(np.split(K, H, axis=-1), axis=0)                   # :74
    idx_ = np.concatenate([ei + i * N for i in range(H)], axis=1)          # :76
    scale = np.sqrt(np.float32(Q_.shape[-1]))                              # :78
    score_ = ((Q_.astype(acc) * K_.astype(acc)).sum(-1) / scale).astype(np.float32)  # :79
    N_ = N * H                                                             # :82
    att_ = segment_softmax(score_, idx_[0], N_, acc=acc)                   # :83-84
    V_ = np.concatenate(np.split(V, H, axis=-1), axis=0)                   # :87
    h_ = spmm(idx_, att_, (N_, N_), V_, acc=acc)                           # :89
    if split_value_heads:
        h = np.concatenate(np.split(h_, H, axis=0), axis=-1)               # :112
    else:
        h = (np.add.reduce(np.stack(np.split(h_.astype(acc), H, axis=0)), axis=0) / H).astype(np.float32)  # :114
    if bias is not None:
        h = h + np.asarray(bias, np.float32)                               # :116-117
    return _act(activation, h).astype(np.float32)                          # :119-120


def _sage_combine(from_x, from_neigh, bias, activation, concat, normalize):
    h = np.concatenate([from_x, from_neigh], axis=1) if concat else from_x + from_neigh
    if bias is not None:
        h = h + np.asarray(bias, np.float32)
    h = _act(activation, h)
    if normalize:
        h = l2_normalize(h)
    return h.astype(np.float32)


def _sage_reduce(x, edge_index, edge_weight, reducer, acc):
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    neighbor_x = x[col]
    if edge_weight is not None:
        neighbor_x = gcn_mapper(None, neighbor_x, edge_weight)
    return reducer(neighbor_x, row, num_nodes=N, acc=acc)


def mean_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias=None, activation=None,
                    concat=True, normalize=False, acc=np.float64):
    """nn/conv/graph_sage.py:9-60."""
    red = _sage_reduce(x, edge_index, edge_weight, mean_reducer, acc)      # :34-41
    return _sage_combine(matmul(x, self_kernel, acc), matmul(red, neighbor_kernel, acc),
                         bias, activation, concat, normalize)              # :43-58


def sum_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_kernel, bias=None, activation=None,
                   concat=True, normalize=False, acc=np.float64):
    """nn/conv/graph_sage.py:64-115."""
    red = _sage_reduce(x, edge_index, edge_weight, sum_reducer, acc)       # :89-96
    return _sage_combine(matmul(x, self_kernel, acc), matmul(red, neighbor_kernel, acc),
                         bias, activation, concat, normalize)


def gcn_graph_sage(x, edge_index, edge_weight, kernel, bias=None, activation=None, normalize=False,
                   cache=None, acc=np.float64):
    """nn/conv/graph_sage.py:118-161 incl. its quirks: a given edge_weight is replaced by ones (:139-140)
    and `cache` lands in gcn_norm_edge's `renorm` slot (:142 vs gcn.py:180): cache=None -> renorm=False
    (falsy), a non-empty dict -> renorm=True, an EMPTY dict -> falsy -> renorm=False."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    E = np.asarray(edge_index).shape[1]
    if edge_weight is not None:
        edge_weight = np.ones([E], np.float32)
    renorm = bool(cache)
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, renorm=renorm, improved=False, acc=acc)
    red = spmm(ei, nw, (N, N), x, acc=acc)                                 # :143-150
    h = matmul(red, kernel, acc)                                           # :152
    if bias is not None:
        h = h + np.asarray(bias, np.float32)
    h = _act(activation, h)
    if normalize:
        h = l2_normalize(h)
    return h.astype(np.float32)


def _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                     neighbor_mlp_bias, bias, activation, concat, normalize, reducer, acc):
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    E = np.asarray(edge_index).shape[1]
    if edge_weight is not None:
        edge_weight = np.ones([E], np.float32)                             # :190-191 / :253-254
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    neighbor_x = gcn_mapper(None, x[col], edge_weight)                     # :197 / :260 (None -> raises)
    h = matmul(neighbor_x, neighbor_mlp_kernel, acc)                       # per-EDGE GEMM, as the reference
    if neighbor_mlp_bias is not None:
        h = h + np.asarray(neighbor_mlp_bias, np.float32)
    h = _act(activation, h)
    red = reducer(h.astype(np.float32), row, num_nodes=N, acc=acc)
    return _sage_combine(matmul(x, self_kernel, acc), matmul(red, neighbor_kernel, acc),
                         bias, activation, concat, normalize)


def mean_pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                         neighbor_mlp_bias=None, bias=None, activation=None, concat=True, normalize=False,
                         acc=np.float64):
    """nn/conv/graph_sage.py:164-225."""
    return _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                            neighbor_mlp_bias, bias, activation, concat, normalize, mean_reducer, acc)


def max_pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                        neighbor_mlp_bias=None, bias=None, activation=None, concat=True, normalize=False,
                        acc=np.float64):
    """nn/conv/graph_sage.py:228-287."""
    return _pool_graph_sage(x, edge_index, edge_weight, self_kernel, neighbor_mlp_kernel, neighbor_kernel,
                            neighbor_mlp_bias, bias, activation, concat, normalize, max_reducer, acc)


# ----------------------------------------------------------------------------
# synthetic inputs — shared by tests, smoke and bench
# ----------------------------------------------------------------------------

def synthetic_edges(num_nodes, num_edges, seed=0):
    """E/2 uniform pairs, self pairs dropped, both directions emitted in the reference's to_directed order
    (all (a,b) then all (b,a), utils/graph_utils.py:186-190); duplicates allowed (they sum)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    half = num_edges // 2
    a = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    b = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    keep = a != b
    a, b = a[keep], b[keep]
    return np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)


def glorot_uniform(rng, fan_in, fan_out):
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=(fan_in, fan_out)).astype(np.float32)


# ----------------------------------------------------------------------------
# "next row 2" convolutions: restated line by line for the parity tests
# ----------------------------------------------------------------------------

def _mlp_encoder(x, kernels, biases, dense_activation, acc):
    h = np.asarray(x, np.float32)
    if kernels is not None:
        last = len(kernels) - 1
        for i, (k_, b_) in enumerate(zip(kernels, biases)):
            h = matmul(h, k_, acc)
            if b_ is not None:
                h = h + np.asarray(b_, np.float32)
            if i < last:
                h = _act(dense_activation, h)
    return h.astype(np.float32)


def gin(x, edge_index, mlp_model, eps=0.0, acc=np.float64):
    """nn/conv/gin.py:11-38."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    E = np.asarray(edge_index).shape[1]
    neighbor_h = spmm(edge_index, np.ones(E, np.float32), (N, N), x, acc=acc)     # :32-34
    return mlp_model((x * (1.0 + eps) + neighbor_h).astype(np.float32))           # :35-36


def sgc(x, edge_index, edge_weight, k, kernel, bias=None, activation=None, renorm=True, improved=False, acc=np.float64):
    """nn/conv/sgc.py:10-61."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, renorm=renorm, improved=improved, acc=acc)
    h = matmul(x, kernel, acc)
    for _ in range(k):
        h = spmm(ei, nw, (N, N), h, acc=acc)
    if bias is not None:
        h = h + np.asarray(bias, np.float32)
    return _act(activation, h).astype(np.float32)


def tagcn(x, edge_index, edge_weight, k, kernel, bias=None, activation=None, renorm=False, improved=False, acc=np.float64):
    """nn/conv/tagcn.py:10-51."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, renorm=renorm, improved=improved, acc=acc)
    xs = [x]
    for _ in range(k):
        xs.append(spmm(ei, nw, (N, N), xs[-1], acc=acc))
    out = matmul(np.concatenate(xs, axis=-1), kernel, acc)
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return _act(activation, out).astype(np.float32)


def appnp(x, edge_index, edge_weight, kernels, biases, dense_activation="relu", activation=None, k=10, alpha=0.1,
          acc=np.float64):
    """nn/conv/appnp.py:11-92 (inference)."""
    N = np.asarray(x).shape[0]
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, acc=acc)
    h = _mlp_encoder(x, kernels, biases, dense_activation, acc)
    output = h
    for _ in range(k):
        output = spmm(ei, nw, (N, N), output, acc=acc)
        output = (output * (1.0 - alpha) + h * alpha).astype(np.float32)
    return _act(activation, output).astype(np.float32)


def ssgc(x, edge_index, edge_weight, kernels=None, biases=None, k=10, alpha=0.1, dense_activation="relu",
         activation=None, acc=np.float64):
    """nn/conv/ssgc.py:11-99 (inference)."""
    N = np.asarray(x).shape[0]
    ei, nw = gcn_norm_adj(edge_index, edge_weight, N, acc=acc)
    h = _mlp_encoder(x, kernels, biases, dense_activation, acc)
    output = h * alpha
    for _ in range(k):
        h = spmm(ei, nw, (N, N), h, acc=acc)
        output = output + (1 - alpha) * h / k
    return _act(activation, output).astype(np.float32)


def le_conv(x, edge_index, edge_weight, self_kernel, self_bias, aggr_self_kernel, aggr_self_bias,
            aggr_neighbor_kernel, aggr_neighbor_bias, activation=None, acc=np.float64):
    """nn/conv/le_conv.py:5-52, incl. its indexing: both gathered terms use `col` (:40-41)."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    ei = np.asarray(edge_index)
    w = np.ones(ei.shape[1], np.float32) if edge_weight is None else np.asarray(edge_weight, np.float32)

    def lin(k_, b_):
        h = matmul(x, k_, acc)
        return h + np.asarray(b_, np.float32) if b_ is not None else h
    self_h, a_self, a_nb = lin(self_kernel, self_bias), lin(aggr_self_kernel, aggr_self_bias), lin(aggr_neighbor_kernel, aggr_neighbor_bias)
    row, col = ei[0], ei[1]
    rep = (a_self[col].astype(acc) - a_nb[col].astype(acc)) * w[:, None].astype(acc)
    aggr = unsorted_segment_sum(rep, row, N, acc=acc).astype(np.float32)
    return _act(activation, self_h + aggr).astype(np.float32)


def get_laplacian(edge_index, num_nodes, edge_weight, normalization_type, fill_weight=1.0, acc=np.float64):
    """utils/graph_utils.py:554-603."""
    ei = np.asarray(edge_index, np.int32)
    w = np.asarray(edge_weight, np.float32)
    row, col = ei[0], ei[1]
    deg = unsorted_segment_sum(w.astype(acc), row, num_nodes, acc=acc)
    with np.errstate(divide="ignore", invalid="ignore"):
        if normalization_type is None:
            ei2, w2 = add_self_loop_edge(ei, num_nodes, w, fill_weight=fill_weight)
            return ei2, (_remove_inf_and_nan(deg)[ei2[0]] - w2).astype(np.float32)
        if normalization_type == "sym":
            dis = _remove_inf_and_nan(np.power(deg, -0.5))
            nw = dis[row] * w * dis[col]
        else:
            dinv = _remove_inf_and_nan(1.0 / deg)
            nw = dinv[row] * w
    return add_self_loop_edge(ei, num_nodes, nw.astype(np.float32), fill_weight=fill_weight)


def laplacian_max_eigenvalue(edge_index, num_nodes, edge_weight, normalization_type):
    """utils/graph_utils.py:884-909 (scipy eigsh / eigs, k=1, which='LM'), restated with a DENSE eigen-solve:
    the largest-magnitude eigenvalue of get_laplacian's matrix.  ('rw' is similar to 'sym': same spectrum.)"""
    nt = "sym" if normalization_type == "rw" else normalization_type
    lei, lw = get_laplacian(edge_index, num_nodes, edge_weight, nt)
    M = np.zeros((num_nodes, num_nodes), np.float64)
    np.add.at(M, (lei[0], lei[1]), lw.astype(np.float64))
    # 'sym' is symmetric on an undirected graph; None yields (deg_r - w_rc) off the diagonal — NOT symmetric, which is why
    # the reference switches from eigsh to eigs there (graph_utils.py:901-905): general dense solve, real part
    ev = np.linalg.eigvals(M)
    return float(ev[np.argmax(np.abs(ev))].real)


def chebynet(x, edge_index, edge_weight, k, kernels, bias=None, activation=None, normalization_type="sym",
             acc=np.float64, use_dynamic_lambda_max=False):
    """nn/conv/chebynet.py:27-137; lambda_max = 2.0 unless use_dynamic_lambda_max (:39-40)."""
    x = np.asarray(x, np.float32)
    N = x.shape[0]
    ei = np.asarray(edge_index, np.int32)
    w = np.ones(ei.shape[1], np.float32) if edge_weight is None else np.asarray(edge_weight, np.float32)
    keep = ei[0] != ei[1]                                               # remove_self_loop_edge (:34)
    lei, lw = get_laplacian(ei[:, keep], N, w[keep], normalization_type, acc=acc)
    lambda_max = laplacian_max_eigenvalue(ei[:, keep], N, w[keep], normalization_type) if use_dynamic_lambda_max else 2.0
    lw = ((2.0 * lw) / lambda_max).astype(np.float32)                    # :43
    T0 = x
    out = matmul(T0, kernels[0], acc)
    if k > 1:
        T1 = spmm(lei, lw, (N, N), x, acc=acc)
        out = out + matmul(T1, kernels[1], acc)
    for i in range(2, k):
        T2 = (spmm(lei, lw, (N, N), T1, acc=acc) * 2.0 - T0).astype(np.float32)
        out = out + matmul(T2, kernels[i], acc)
        T0, T1 = T1, T2
    if bias is not None:
        out = out + np.asarray(bias, np.float32)
    return _act(activation, out).astype(np.float32)


# ----------------------------------------------------------------------------
# pooling: top-k selection
# ----------------------------------------------------------------------------

def topk_pool(source_index, score, k=None, ratio=None):
    """nn/pool/topk_pool.py:6-87, step by step (dense padded score matrix, row-wise argsort, mask).
    tf.argsort is top_k underneath, so both sorts are stable with the lower index first on ties; a ratio whose
    node_k exceeds a source's count would select the reference's padding columns — restated as "keep all"."""
    if k is None and ratio is None:
        raise Exception("you should provide either k or ratio for topk_pool")
    elif k is not None and ratio is not None:
        raise Exception("you should provide either k or ratio for topk_pool, not both of them")
    source_index = np.asarray(source_index, dtype=np.int64).reshape(-1)
    score = np.asarray(score, dtype=np.float32).reshape(-1)
    if source_index.shape[0] == 0:
        return np.zeros(0, dtype=np.int32)
    perm = np.argsort(source_index, kind="stable")                               # :31
    sorted_source = source_index[perm]
    sorted_score = score[perm]
    num_targets = sorted_source.shape[0]
    counts = np.bincount(sorted_source)                                          # :38-39 segment_sum of ones
    num_cols = int(counts.max())
    num_seen = counts.shape[0]
    min_score = sorted_score.min()
    before = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)      # :48-51
    target_in_source = np.arange(num_targets) - before[sorted_source]            # :53
    score_matrix = np.full((num_seen, num_cols), min_score - np.float32(1.0), dtype=np.float32)   # :55
    score_matrix[sorted_source, target_in_source] = sorted_score                 # :56-57
    sort_index = np.argsort(-score_matrix, axis=-1, kind="stable")               # :59 DESCENDING
    if k is not None:
        node_k = np.minimum(k, counts)                                           # :62-65
    else:
        node_k = np.ceil(counts.astype(np.float32) * np.float32(ratio)).astype(np.int64)   # :67-70
        node_k = np.minimum(node_k, counts)
    out = []
    for s in range(num_seen):                                                    # :73-84 meshgrid + mask, row-major
        cols = sort_index[s, :node_k[s]]
        out.append(before[s] + cols)
    topk_index = np.concatenate(out) if out else np.zeros(0, np.int64)
    return perm[topk_index].astype(np.int32)                                     # :89


def neighbor_lists(edge_index, edge_weight=None, sampled_node_index=None):
    """The deterministic part of RandomNeighborSampler (utils/graph_utils.py:631-731): per sampled row, the list of
    (virtual neighbour id, weight) the random draw chooses from, in the reference's order.  Returns
    (virtual_row_ids, [ (neighbour ids, weights), ... ]) skipping rows without (kept) neighbours, as the loop does."""
    edge_index = np.asarray(edge_index, dtype=np.int64)
    w = np.ones(edge_index.shape[1], np.float32) if edge_weight is None else np.asarray(edge_weight, np.float32)
    num_rows, num_cols = int(edge_index[0].max()) + 1, int(edge_index[1].max()) + 1
    nbr = {}
    for (a, b), ww in zip(edge_index.T, w):                                           # :645-655
        nbr.setdefault(int(a), ([], []))
        nbr[int(a)][0].append(int(b))
        nbr[int(a)][1].append(ww)
    if sampled_node_index is None:
        rows = sorted(nbr.keys())                                                     # :662
        virt = rows
        col_map = None
    else:
        if isinstance(sampled_node_index, tuple):
            rows, cols = sampled_node_index
        else:
            rows = cols = sampled_node_index
        rows, cols = [int(r) for r in rows], [int(c) for c in cols]
        virt = list(range(len(rows)))
        col_map = -np.ones(num_cols, np.int64)                                        # :701-703
        for i, c in enumerate(cols):
            if c < num_cols:
                col_map[c] = i
    out_rows, out = [], []
    for vi, r in zip(virt, rows):                                                     # :716-731
        if r not in nbr:
            continue
        ids, ws = np.asarray(nbr[r][0]), np.asarray(nbr[r][1], np.float32)
        if col_map is not None:
            v = col_map[ids]
            m = v >= 0
            ids, ws = v[m], ws[m]
            if len(ids) == 0:
                continue
        out_rows.append(vi)
        out.append((ids, ws))
    return out_rows, out
