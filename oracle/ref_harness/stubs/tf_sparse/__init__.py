# coding=utf-8
"""
A stand-in for ``tf_sparse`` (CrawlScript/tf_sparse, the reference pins ``tf_sparse >= 0.0.17``, setup.py:25) — TEST
INFRASTRUCTURE ONLY, see ../tensorflow/__init__.py for why it exists.

tf_sparse is a third-party dependency of the reference that is absent from /root/reference and not installable
here, so this file RESTATES the part of its published behaviour the reference's call sites rely on; every method is
written over the ``tensorflow`` stand-in's primitives (gather / unsorted_segment_* / unique), i.e. exactly the
formulation tf_sparse itself uses, so results carry TF-CPU float32 semantics:

  SparseMatrix(index, value=None, shape=None)  COO matrix: index int32 [2, nnz], value float32 [nnz] (default ones),
                                               shape defaults to [max(index)+1]*2; ``.index/.value/._shape`` are
                                               tensors (the reference calls ``.numpy()`` on all three, gcn.py:128)
  A @ H / A.matmul(H, num_or_size_splits)      out[r] = sum_{e: row_e = r} value_e * H[col_e]; duplicates are summed;
                                               column splits of H are multiplied one by one and concatenated
  D @ A, A @ D with D = tfs.diags(v)           value_e * v[row_e]   /   value_e * v[col_e], index unchanged
  A.segment_sum(axis=-1 | 0)                   row sums | column sums of the values
  A.segment_softmax(axis=-1)                   the reference's own nn/kernel/segment.py:26-33 formula, by row
  A.add_diag(c)                                A + c*I: the diagonal COO entries are appended and duplicated
                                               coordinates are merged by summation in first-occurrence order
                                               (tf.unique over the row*ncols+col hash), so an existing (i,i) entry is
                                               added to, never replaced
  A.dropout(rate, training)                    identity unless ``training`` and rate > 0, else tf.nn.dropout(value)
  A.transpose(), A.to_dense(), A + B, tfs.shape, tfs.diags, tfs.eye, tfs.concat
"""
import numpy as np
import tensorflow as tf


def _merge_duplicated(index, value, num_cols):
    """Sum the values of duplicated coordinates, keeping first-occurrence order."""
    edge_hash = tf.cast(index[0], tf.int64) * tf.cast(num_cols, tf.int64) + tf.cast(index[1], tf.int64)
    unique_hash, unique_index = tf.unique(edge_hash)
    n_unique = tf.shape(unique_hash)[0]
    row = tf.cast(tf.math.floordiv(unique_hash, tf.cast(num_cols, tf.int64)), tf.int32)
    col = tf.cast(tf.math.floormod(unique_hash, tf.cast(num_cols, tf.int64)), tf.int32)
    merged_value = tf.math.unsorted_segment_sum(value, unique_index, n_unique)
    return tf.stack([row, col], axis=0), merged_value


class SparseMatrix(object):
    __array_priority__ = 10000
    __array_ufunc__ = None          # numpy arrays defer `x @ A` to A.__rmatmul__

    def __init__(self, index, value=None, shape=None, merge=False):
        self.index = tf.cast(tf.convert_to_tensor(index), tf.int32)
        if value is None:
            self.value = tf.ones([tf.shape(self.index)[1]], dtype=tf.float32)
        else:
            value = tf.convert_to_tensor(value)
            if value.dtype == tf.float64:
                value = tf.cast(value, tf.float32)
            self.value = value
        if shape is None:
            n = tf.reduce_max(self.index) + 1
            shape = [n, n]
        self._shape = tf.cast(tf.convert_to_tensor([int(s) for s in shape]), tf.int32)
        if merge:
            self.index, self.value = _merge_duplicated(self.index, self.value, self._shape[1])

    # ---- structure -------------------------------------------------------------------------------------------
    @property
    def row(self):
        return self.index[0]

    @property
    def col(self):
        return self.index[1]

    @property
    def shape(self):
        return [int(self._shape[0]), int(self._shape[1])]

    @property
    def dtype(self):
        return self.value.dtype

    def with_value(self, value):
        return self.__class__(self.index, value, self._shape)

    def merge_duplicated_index(self):
        index, value = _merge_duplicated(self.index, self.value, self._shape[1])
        return self.__class__(index, value, self._shape)

    def transpose(self):
        return self.__class__(tf.stack([self.col, self.row], axis=0), self.value, [self._shape[1], self._shape[0]])

    def to_dense(self):
        return tf.scatter_nd(tf.transpose(self.index), self.value, self._shape)

    def to_sparse_tensor(self):
        return tf.sparse.reorder(tf.sparse.SparseTensor(tf.cast(tf.transpose(self.index), tf.int64), self.value,
                                                        tf.cast(self._shape, tf.int64)))

    @classmethod
    def from_sparse_tensor(cls, sparse_tensor):
        return cls(tf.transpose(sparse_tensor.indices), sparse_tensor.values, sparse_tensor.dense_shape)

    # ---- reductions over the stored values -------------------------------------------------------------------
    def segment_sum(self, axis=-1, keepdims=False):
        if axis in (-1, 1):
            out = tf.math.unsorted_segment_sum(self.value, self.row, self._shape[0])
        elif axis in (0, -2):
            out = tf.math.unsorted_segment_sum(self.value, self.col, self._shape[1])
        else:
            raise Exception("invalid axis: {}".format(axis))
        return tf.expand_dims(out, axis) if keepdims else out

    reduce_sum = segment_sum

    def segment_softmax(self, axis=-1):
        if axis in (-1, 1):
            ids, n = self.row, self._shape[0]
        elif axis in (0, -2):
            ids, n = self.col, self._shape[1]
        else:
            raise Exception("invalid axis: {}".format(axis))
        # the formula of the reference's own segment_softmax (nn/kernel/segment.py:26-33)
        max_values = tf.math.unsorted_segment_max(self.value, ids, n)
        exp = tf.exp(self.value - tf.stop_gradient(tf.gather(max_values, ids)))
        denominator = tf.math.unsorted_segment_sum(exp, ids, n) + 1e-8
        return self.with_value(exp / tf.gather(denominator, ids))

    softmax = segment_softmax

    def dropout(self, drop_rate, training=False):
        if training and drop_rate > 0.0:
            return self.with_value(tf.compat.v2.nn.dropout(self.value, drop_rate))
        return self

    # ---- algebra ---------------------------------------------------------------------------------------------
    def add_diag(self, diagonal):
        n = min(self.shape)
        if np.ndim(diagonal) == 0:
            diagonal = tf.cast(tf.fill([n], diagonal), self.value.dtype)
        return self + diags(diagonal, shape=self._shape)

    def __add__(self, other):
        if isinstance(other, SparseMatrix):
            index = tf.concat([self.index, other.index], axis=1)
            value = tf.concat([self.value, tf.cast(other.value, self.value.dtype)], axis=0)
            return SparseMatrix(index, value, self._shape, merge=True)
        return self.to_dense() + other

    __radd__ = __add__

    def __neg__(self):
        return self.with_value(-self.value)

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, scalar):
        return self.with_value(self.value * scalar)

    __rmul__ = __mul__

    def __truediv__(self, scalar):
        return self.with_value(self.value / scalar)

    def matmul_diag(self, diagonal):
        return self.with_value(self.value * tf.gather(diagonal, self.col))

    def rmatmul_diag(self, diagonal):
        return self.with_value(tf.gather(diagonal, self.row) * self.value)

    def matmul_dense(self, h):
        h = tf.convert_to_tensor(h)
        msg = tf.gather(h, self.col) * tf.expand_dims(self.value, -1)
        return tf.math.unsorted_segment_sum(msg, self.row, self._shape[0])

    def matmul(self, h, num_or_size_splits=None):
        if isinstance(h, DiagMatrix):
            return self.matmul_diag(h.diagonal)
        if isinstance(h, SparseMatrix):
            return SparseMatrix.from_dense(self.to_dense() @ h.to_dense())
        if isinstance(h, tf.sparse.SparseTensor):
            h = tf.sparse.to_dense(h)
        if num_or_size_splits is None:
            return self.matmul_dense(h)
        parts = tf.split(h, num_or_size_splits, axis=-1)
        return tf.concat([self.matmul_dense(p) for p in parts], axis=-1)

    def __matmul__(self, h):
        return self.matmul(h)

    def rmatmul_dense(self, h):
        return tf.transpose(self.transpose().matmul_dense(tf.transpose(tf.convert_to_tensor(h))))

    def __rmatmul__(self, h):
        if isinstance(h, DiagMatrix):
            return self.rmatmul_diag(h.diagonal)
        return self.rmatmul_dense(h)

    @classmethod
    def from_dense(cls, dense):
        dense = np.asarray(dense)
        r, c = np.nonzero(dense)
        return cls(np.stack([r, c]).astype(np.int32), dense[r, c], dense.shape)

    def eliminate_zeros(self):
        mask = tf.not_equal(self.value, 0.0)
        return self.__class__(tf.boolean_mask(self.index, mask, axis=1), tf.boolean_mask(self.value, mask),
                              self._shape)

    def __repr__(self):
        return "SparseMatrix(index={}, value={}, shape={})".format(self.index, self.value, self.shape)


class DiagMatrix(SparseMatrix):
    """What ``tfs.diags`` returns: a SparseMatrix that multiplies as a diagonal scaling (no index arithmetic)."""

    def __init__(self, diagonal, shape=None):
        diagonal = tf.convert_to_tensor(diagonal)
        n = int(tf.shape(diagonal)[0])
        r = tf.range(n, dtype=tf.int32)
        SparseMatrix.__init__(self, tf.stack([r, r], axis=0), diagonal, [n, n] if shape is None else shape)
        self.diagonal = self.value

    def with_value(self, value):
        return DiagMatrix(value, self._shape)

    def matmul(self, h, num_or_size_splits=None):
        if isinstance(h, DiagMatrix):
            return DiagMatrix(self.diagonal * h.diagonal, self._shape)
        if isinstance(h, SparseMatrix):
            return h.rmatmul_diag(self.diagonal)
        return tf.expand_dims(self.diagonal, -1) * tf.convert_to_tensor(h)

    def __matmul__(self, h):
        return self.matmul(h)

    def __rmatmul__(self, h):
        if isinstance(h, SparseMatrix):
            return h.matmul_diag(self.diagonal)
        return tf.convert_to_tensor(h) * tf.expand_dims(self.diagonal, 0)


def diags(diagonals, shape=None):
    return DiagMatrix(diagonals, shape)


def eye(num_rows, dtype=tf.float32):
    return DiagMatrix(tf.ones([num_rows], dtype=dtype))


def shape(x):
    if isinstance(x, SparseMatrix):
        return x._shape
    return tf.shape(x)


def sparse_diag_matmul(sparse, diagonal):
    return sparse.matmul_diag(diagonal)


def diag_sparse_matmul(diagonal, sparse):
    return sparse.rmatmul_diag(diagonal)


def concat(sparse_matrices, axis=0):
    axis = axis % 2
    off, idx, val = 0, [], []
    shp = list(sparse_matrices[0].shape)
    for m in sparse_matrices:
        i = np.array(m.index)
        i[axis] += off
        off += m.shape[axis]
        idx.append(i)
        val.append(np.asarray(m.value))
    shp[axis] = off
    return SparseMatrix(np.concatenate(idx, axis=1), np.concatenate(val, axis=0), shp)
