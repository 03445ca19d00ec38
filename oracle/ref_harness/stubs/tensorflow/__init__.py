# coding=utf-8
"""
A numpy-backed stand-in for the ``tensorflow`` module — TEST INFRASTRUCTURE ONLY.

Purpose: let the reference's own, UNMODIFIED Python (``/root/reference/tf_geometric``) be imported and executed in
this image, where TensorFlow is not installable, so that the reference itself generates the golden vectors
(``tests/golden/make_golden_from_reference.py``) and checks ``oracle/tfg_oracle.py``
(``tests/test_oracle_vs_reference.py``).  What runs unmodified: every line of tf_geometric's composition logic
(aggregate_neighbors, segment_softmax, gcn_norm_adj, gcn, gat, the five GraphSAGE variants, graph_utils,
the tfg.layers classes ...).  What is RESTATED here: the ~110 TensorFlow primitives those lines call, each with the
semantics TensorFlow documents for its CPU kernels (eager mode, float32 arithmetic kept in float32):

  tf.gather                      -> numpy take along ``axis``
  tf.math.unsorted_segment_sum   -> zeros, then ``out[ids[i]] += data[i]`` in index order (np.add.at), fp32
  tf.math.unsorted_segment_mean  -> segment_sum / max(count, 1)
  tf.math.unsorted_segment_max   -> initial value ``dtype.lowest`` (so an empty segment is float32 lowest)
  tf.math.unsorted_segment_min   -> initial value ``dtype.max``
  tf.nn.l2_normalize             -> x * rsqrt(max(sum(x*x), 1e-12))
  tf.unique                      -> unique values in FIRST-OCCURRENCE order + index of each element
  tf.nn.dropout                  -> identity when rate == 0, else mask/(1-rate) from tf.random's numpy generator
  everything else                -> the numpy function of the same meaning

This is not a TensorFlow re-implementation: no graph mode, no gradients, no devices.  ``tf.function`` returns the
python function itself.
"""
import numpy as _np

__version__ = "2.4.1"      # the version the reference's docs pin (doc/requirements.txt:5); selects the TF2 branches

# ---------------------------------------------------------------------------------------------------------------
# dtypes: numpy dtypes (np.dtype('float32') == np.float32 is True, which is all the reference compares)
# ---------------------------------------------------------------------------------------------------------------
float16 = _np.dtype("float16")
float32 = _np.dtype("float32")
float64 = _np.dtype("float64")
int8 = _np.dtype("int8")
int16 = _np.dtype("int16")
int32 = _np.dtype("int32")
int64 = _np.dtype("int64")
uint8 = _np.dtype("uint8")
bool = _np.dtype("bool")       # noqa: A001  (tf.bool)
string = _np.dtype("O")

_pybool = (1 == 1).__class__


class Tensor(_np.ndarray):
    """An eager tensor: a numpy array that answers ``.numpy()`` and for which ``tf.is_tensor`` is True."""

    def numpy(self):
        a = _np.asarray(self)
        return a[()] if a.ndim == 0 else a

    def __getitem__(self, item):
        if isinstance(item, Tensor) and item.dtype == _np.bool_ and item.ndim == 0:
            item = _pybool(item)
        out = _np.ndarray.__getitem__(self, _np.asarray(item) if isinstance(item, Tensor) else item)
        return _t(out)

    def __hash__(self):
        return id(self)

    def __bool__(self):
        return _pybool(_np.asarray(self).all()) if self.size == 1 else _np.ndarray.__bool__(self)

    def __matmul__(self, other):
        if hasattr(other, "__rmatmul__") and not isinstance(other, _np.ndarray):
            r = other.__rmatmul__(self)
            if r is not NotImplemented:
                return r
        return matmul(self, other)

    def __rmatmul__(self, other):
        return matmul(other, self)

    def get_shape(self):
        return TensorShape(self.shape)

    def __repr__(self):
        return "tf_stub.Tensor({}, shape={}, dtype={})".format(_np.asarray(self), self.shape, self.dtype)


class Variable(Tensor):
    def __new__(cls, initial_value, trainable=True, name=None, dtype=None):
        a = _np.array(initial_value, dtype=dtype, copy=True)
        obj = a.view(cls)
        obj._name = name
        obj.trainable = trainable
        return obj

    def __array_finalize__(self, obj):
        self._name = getattr(obj, "_name", None)
        self.trainable = getattr(obj, "trainable", True)

    @property
    def name(self):
        return self._name

    def assign(self, value):
        _np.asarray(self)[...] = _np.asarray(value, dtype=self.dtype)
        return self

    def value(self):
        return _t(_np.asarray(self))


class TensorShape(tuple):
    def as_list(self):
        return list(self)


class TensorSpec(object):
    def __init__(self, shape=None, dtype=float32, name=None):
        self.shape, self.dtype, self.name = shape, dtype, name


def _t(x, dtype=None):
    """Wraps a numpy value as a Tensor (python floats become float32, python ints int32, as tf.convert_to_tensor)."""
    if isinstance(x, (SparseTensor,)):
        return x
    if isinstance(x, Tensor) and (dtype is None or x.dtype == dtype) and not isinstance(x, Variable):
        return x
    if dtype is None:
        if isinstance(x, _pybool):
            dtype = _np.bool_
        elif isinstance(x, int):
            dtype = _np.int32
        elif isinstance(x, float):
            dtype = _np.float32
        elif isinstance(x, (list, tuple)):
            a = _np.asarray([_np.asarray(v) for v in x]) if len(x) and any(isinstance(v, _np.ndarray) for v in x) \
                else _np.asarray(x)
            if a.dtype == _np.float64 and not _has_np_float64(x):
                a = a.astype(_np.float32)
            elif a.dtype == _np.int64 and not _has_np_int64(x):
                a = a.astype(_np.int32)
            return a.view(Tensor)
    a = _np.asarray(x, dtype=dtype)
    return a.view(Tensor)


def _has_np_float64(x):
    if isinstance(x, (list, tuple)):
        return any(_has_np_float64(v) for v in x)
    return isinstance(x, (_np.ndarray, _np.generic)) and _np.asarray(x).dtype == _np.float64


def _has_np_int64(x):
    if isinstance(x, (list, tuple)):
        return any(_has_np_int64(v) for v in x)
    return isinstance(x, (_np.ndarray, _np.generic)) and _np.asarray(x).dtype == _np.int64


def _a(x):
    """tensor-like -> plain ndarray (python scalars keep being weak scalars for numpy's promotion)."""
    if isinstance(x, (int, float, _pybool)):
        return x
    if isinstance(x, (list, tuple)):
        return _np.asarray(_t(x))
    return _np.asarray(x)


# ---------------------------------------------------------------------------------------------------------------
# basics
# ---------------------------------------------------------------------------------------------------------------
def is_tensor(x):
    return isinstance(x, (Tensor, SparseTensor))


def executing_eagerly():
    return True


def enable_eager_execution():
    return None


def function(func=None, *args, **kwargs):
    if func is not None and callable(func):
        return func

    def decorate(f):
        return f
    return decorate


def convert_to_tensor(value, dtype=None, name=None):
    return _t(value, dtype)


constant = convert_to_tensor


def identity(x):
    return _t(_np.array(_a(x)))


def stop_gradient(x):
    return _t(x)


def cast(x, dtype):
    if isinstance(x, SparseTensor):
        return SparseTensor(x.indices, cast(x.values, dtype), x.dense_shape)
    return _t(_np.asarray(_a(x)).astype(dtype))


def shape(x, out_type=int32):
    if isinstance(x, SparseTensor):
        return _t(_np.asarray(x.dense_shape, dtype=out_type))
    return _t(_np.asarray(_np.shape(_a(x)), dtype=out_type))


def size(x):
    return _t(_np.int32(_np.size(_a(x))))


def rank(x):
    return _t(_np.int32(_np.ndim(_a(x))))


def reshape(x, shape):      # noqa: A002
    return _t(_np.reshape(_a(x), [int(s) for s in _np.asarray(_a(shape)).reshape(-1)]))


def expand_dims(x, axis):
    return _t(_np.expand_dims(_a(x), axis))


def squeeze(x, axis=None):
    return _t(_np.squeeze(_a(x), axis=axis))


def transpose(x, perm=None):
    return _t(_np.transpose(_a(x), perm))


def concat(values, axis):
    return _t(_np.concatenate([_np.asarray(_a(v)) for v in values], axis=int(axis)))


def stack(values, axis=0):
    return _t(_np.stack([_np.asarray(_a(v)) for v in values], axis=axis))


def unstack(x, axis=0):
    return [_t(v) for v in _np.moveaxis(_a(x), axis, 0)]


def split(value, num_or_size_splits, axis=0):
    v = _np.asarray(_a(value))
    if isinstance(num_or_size_splits, (int, _np.integer)) or _np.ndim(num_or_size_splits) == 0:
        n = int(num_or_size_splits)
        if v.shape[axis] % n != 0:
            raise ValueError("Dimension size must be evenly divisible by {} but is {}".format(n, v.shape[axis]))
        return [_t(p) for p in _np.split(v, n, axis=axis)]
    sizes = [int(s) for s in num_or_size_splits]
    return [_t(p) for p in _np.split(v, _np.cumsum(sizes)[:-1], axis=axis)]


def tile(x, multiples):
    return _t(_np.tile(_a(x), [int(m) for m in _a(multiples)]))


def gather(params, indices, axis=0, batch_dims=0):
    p, i = _np.asarray(_a(params)), _np.asarray(_a(indices))
    if i.size and (i.min() < 0 or i.max() >= p.shape[axis]):
        raise errors.InvalidArgumentError("indices out of range in tf.gather")       # TF-CPU behaviour
    return _t(_np.take(p, i, axis=axis))


def gather_nd(params, indices):
    p, i = _np.asarray(_a(params)), _np.asarray(_a(indices))
    return _t(p[tuple(_np.moveaxis(i, -1, 0))])


def boolean_mask(tensor, mask, axis=None):
    t, m = _np.asarray(_a(tensor)), _np.asarray(_a(mask)).astype(_np.bool_)
    axis = 0 if axis is None else axis
    return _t(_np.compress(m, t, axis=axis))


def where(condition, x=None, y=None):
    c = _np.asarray(_a(condition))
    if x is None and y is None:
        return _t(_np.argwhere(c).astype(_np.int64))
    return _t(_np.where(c, _a(x), _a(y)))


def scatter_nd(indices, updates, shape):        # noqa: A002
    i, u = _np.asarray(_a(indices)), _np.asarray(_a(updates))
    out = _np.zeros([int(s) for s in _a(shape)], dtype=u.dtype)
    _np.add.at(out, tuple(_np.moveaxis(i, -1, 0)), u)
    return _t(out)


def tensor_scatter_nd_update(tensor, indices, updates):
    out = _np.array(_a(tensor))
    i = _np.asarray(_a(indices))
    out[tuple(_np.moveaxis(i, -1, 0))] = _a(updates)
    return _t(out)


def range(start, limit=None, delta=1, dtype=None):      # noqa: A001
    if limit is None:
        start, limit = 0, start
    a = _np.arange(_a(start), _a(limit), _a(delta))
    if dtype is None:
        dtype = _np.int32 if a.dtype.kind == "i" else _np.float32
    return _t(a.astype(dtype))


def _shape_list(shape):     # noqa: A002
    s = _np.asarray(_a(shape)).reshape(-1)
    return [int(v) for v in s]


def ones(shape, dtype=float32):     # noqa: A002
    return _t(_np.ones(_shape_list(shape), dtype=dtype))


def zeros(shape, dtype=float32):    # noqa: A002
    return _t(_np.zeros(_shape_list(shape), dtype=dtype))


def fill(dims, value):
    v = _np.asarray(_t(value))
    return _t(_np.full(_shape_list(dims), v, dtype=v.dtype))


def ones_like(x, dtype=None):
    return _t(_np.ones_like(_a(x), dtype=dtype))


def zeros_like(x, dtype=None):
    return _t(_np.zeros_like(_a(x), dtype=dtype))


def eye(num_rows, num_columns=None, dtype=float32):
    return _t(_np.eye(int(num_rows), None if num_columns is None else int(num_columns), dtype=dtype))


def meshgrid(*args, **kwargs):
    return [_t(m) for m in _np.meshgrid(*[_a(v) for v in args], indexing=kwargs.get("indexing", "xy"))]


def argsort(values, axis=-1, direction="ASCENDING", stable=False):
    v = _np.asarray(_a(values))
    if direction == "DESCENDING":
        # TF implements DESCENDING as an ascending sort of the negated values (stable for ties in index order)
        v = -v.astype(_np.int64) if v.dtype.kind in "iu" else -v
    return _t(_np.argsort(v, axis=axis, kind="stable").astype(_np.int32))


def sort(values, axis=-1, direction="ASCENDING"):
    v = _np.sort(_a(values), axis=axis, kind="stable")
    return _t(_np.flip(v, axis=axis) if direction == "DESCENDING" else v)


def unique(x, out_idx=int32):
    """tf.unique: y in first-occurrence order, idx[i] = position of x[i] in y."""
    v = _np.asarray(_a(x))
    y_sorted, first, inv = _np.unique(v, return_index=True, return_inverse=True)
    order = _np.argsort(first, kind="stable")             # sorted-unique rank -> first-occurrence rank
    rank_of = _np.empty_like(order)
    rank_of[order] = _np.arange(order.size)
    return _t(y_sorted[order]), _t(rank_of[inv.reshape(-1)].astype(out_idx))


def _reduce(fn):
    def op(x, axis=None, keepdims=False):
        if isinstance(axis, (list, tuple)):
            axis = tuple(axis)
        a = _np.asarray(_a(x))
        return _t(fn(a, axis=axis, keepdims=keepdims).astype(a.dtype, copy=False)
                  if fn not in (_np.any, _np.all) else fn(a, axis=axis, keepdims=keepdims))
    return op


reduce_sum = _reduce(_np.sum)
reduce_max = _reduce(_np.max)
reduce_min = _reduce(_np.min)
reduce_mean = _reduce(_np.mean)
reduce_prod = _reduce(_np.prod)
reduce_any = _reduce(_np.any)
reduce_all = _reduce(_np.all)


def _binary(fn):
    def op(x, y, name=None):
        return _t(fn(_a(x), _a(y)))
    return op


def _unary(fn):
    def op(x, name=None):
        return _t(fn(_a(x)))
    return op


add = _binary(_np.add)
subtract = _binary(_np.subtract)
multiply = _binary(_np.multiply)
divide = _binary(_np.true_divide)
maximum = _binary(_np.maximum)
minimum = _binary(_np.minimum)
equal = _binary(_np.equal)
not_equal = _binary(_np.not_equal)
less = _binary(_np.less)
less_equal = _binary(_np.less_equal)
greater = _binary(_np.greater)
greater_equal = _binary(_np.greater_equal)
logical_and = _binary(_np.logical_and)
logical_or = _binary(_np.logical_or)
logical_not = _unary(_np.logical_not)
exp = _unary(_np.exp)
log = _unary(_np.log)
sqrt = _unary(_np.sqrt)
abs = _unary(_np.abs)       # noqa: A001
square = _unary(_np.square)
sign = _unary(_np.sign)
tanh = _unary(_np.tanh)


def pow(x, y):      # noqa: A001
    a = _np.asarray(_a(x))
    with _np.errstate(divide="ignore", invalid="ignore"):
        if a.dtype.kind == "f":
            return _t(_np.power(a, _np.asarray(y, dtype=a.dtype)))
        return _t(_np.power(a, _a(y)))


def add_n(inputs):
    out = _np.array(_a(inputs[0]))
    for v in inputs[1:]:
        out = out + _a(v)
    return _t(out)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _np.asarray(_a(a)), _np.asarray(_a(b))
    if transpose_a:
        a = _np.swapaxes(a, -1, -2)
    if transpose_b:
        b = _np.swapaxes(b, -1, -2)
    return _t(_np.matmul(a, b))


def norm(tensor, ord="euclidean", axis=None, keepdims=False):      # noqa: A002
    a = _np.asarray(_a(tensor))
    if ord in ("euclidean", 2):
        return _t(_np.sqrt(_np.sum(a * a, axis=axis, keepdims=keepdims)).astype(a.dtype))
    if ord == 1:
        return _t(_np.sum(_np.abs(a), axis=axis, keepdims=keepdims))
    raise NotImplementedError(ord)


def cond(pred, true_fn, false_fn):
    return true_fn() if _pybool(_np.asarray(_a(pred))) else false_fn()


def clip_by_value(t, lo, hi):
    return _t(_np.clip(_a(t), _a(lo), _a(hi)))


def one_hot(indices, depth, dtype=float32):
    return _t(_np.eye(int(depth), dtype=dtype)[_np.asarray(_a(indices))])


def argmax(x, axis=None, output_type=int64):
    return _t(_np.argmax(_a(x), axis=axis).astype(output_type))


# ---------------------------------------------------------------------------------------------------------------
# segment ops (the arithmetic the hot path bottoms out in)
# ---------------------------------------------------------------------------------------------------------------
def _seg_args(data, segment_ids, num_segments):
    d = _np.asarray(_a(data))
    ids = _np.asarray(_a(segment_ids)).astype(_np.int64)
    n = int(_np.asarray(_a(num_segments)))
    if ids.size and ids.max() >= n:
        # TF-CPU: InvalidArgumentError "segment_ids[i] = v is out of range [0, n)"; negative ids are dropped
        raise errors.InvalidArgumentError("segment id {} out of range [0, {})".format(int(ids.max()), n))
    keep = ids >= 0
    if not keep.all():
        d, ids = d[keep], ids[keep]
    return d, ids, n


def _unsorted_segment_sum(data, segment_ids, num_segments, name=None):
    d, ids, n = _seg_args(data, segment_ids, num_segments)
    out = _np.zeros((n,) + d.shape[ids.ndim:], dtype=d.dtype)
    _np.add.at(out, ids, d)          # sequential, in index order, in the data's own dtype
    return _t(out)


def _unsorted_segment_mean(data, segment_ids, num_segments, name=None):
    d, ids, n = _seg_args(data, segment_ids, num_segments)
    s = _np.zeros((n,) + d.shape[ids.ndim:], dtype=d.dtype)
    _np.add.at(s, ids, d)
    cnt = _np.maximum(_np.bincount(ids.reshape(-1), minlength=n), 1).astype(d.dtype)
    return _t(s / cnt.reshape((n,) + (1,) * (s.ndim - 1)))


def _lowest(dt):
    return _np.finfo(dt).min if dt.kind == "f" else _np.iinfo(dt).min


def _highest(dt):
    return _np.finfo(dt).max if dt.kind == "f" else _np.iinfo(dt).max


def _unsorted_segment_max(data, segment_ids, num_segments, name=None):
    d, ids, n = _seg_args(data, segment_ids, num_segments)
    out = _np.full((n,) + d.shape[ids.ndim:], _lowest(d.dtype), dtype=d.dtype)
    _np.maximum.at(out, ids, d)
    return _t(out)


def _unsorted_segment_min(data, segment_ids, num_segments, name=None):
    d, ids, n = _seg_args(data, segment_ids, num_segments)
    out = _np.full((n,) + d.shape[ids.ndim:], _highest(d.dtype), dtype=d.dtype)
    _np.minimum.at(out, ids, d)
    return _t(out)


def _sorted_segment(fn_unsorted):
    def op(data, segment_ids, name=None):
        ids = _np.asarray(_a(segment_ids))
        if ids.size and (_np.diff(ids) < 0).any():
            raise errors.InvalidArgumentError("segment ids are not increasing")
        n = int(ids.max()) + 1 if ids.size else 0
        return fn_unsorted(data, ids, n)
    return op


def _segment_max_sorted(data, segment_ids, name=None):
    # tf.math.segment_max: missing ids inside the range yield 0 (not lowest) — documented TF behaviour
    d = _np.asarray(_a(data))
    ids = _np.asarray(_a(segment_ids)).astype(_np.int64)
    n = int(ids.max()) + 1 if ids.size else 0
    out = _np.asarray(_unsorted_segment_max(d, ids, n)).copy()
    present = _np.bincount(ids, minlength=n) > 0
    out[~present] = 0
    return _t(out)


def _segment_min_sorted(data, segment_ids, name=None):
    d = _np.asarray(_a(data))
    ids = _np.asarray(_a(segment_ids)).astype(_np.int64)
    n = int(ids.max()) + 1 if ids.size else 0
    out = _np.asarray(_unsorted_segment_min(d, ids, n)).copy()
    present = _np.bincount(ids, minlength=n) > 0
    out[~present] = 0
    return _t(out)


def _cumsum(x, axis=0, exclusive=False, reverse=False):
    a = _np.asarray(_a(x))
    if reverse:
        a = _np.flip(a, axis)
    c = _np.cumsum(a, axis=axis, dtype=a.dtype)
    if exclusive:
        c = c - a
    if reverse:
        c = _np.flip(c, axis)
    return _t(c)


class _Namespace(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


math = _Namespace(
    unsorted_segment_sum=_unsorted_segment_sum, unsorted_segment_mean=_unsorted_segment_mean,
    unsorted_segment_max=_unsorted_segment_max, unsorted_segment_min=_unsorted_segment_min,
    segment_sum=_sorted_segment(_unsorted_segment_sum), segment_mean=_sorted_segment(_unsorted_segment_mean),
    segment_max=_segment_max_sorted, segment_min=_segment_min_sorted,
    cumsum=_cumsum, is_inf=_unary(_np.isinf), is_nan=_unary(_np.isnan), logical_or=logical_or,
    logical_and=logical_and, logical_not=logical_not, sqrt=sqrt, exp=exp, log=log, pow=pow, abs=abs,
    reduce_min=reduce_min, reduce_max=reduce_max, reduce_sum=reduce_sum, reduce_mean=reduce_mean,
    minimum=minimum, maximum=maximum, floormod=_binary(_np.mod), floordiv=_binary(_np.floor_divide),
    ceil=_unary(_np.ceil), floor=_unary(_np.floor), equal=equal, not_equal=not_equal, greater=greater, less=less,
    add=add, subtract=subtract, multiply=multiply, divide=divide, square=square, tanh=tanh, add_n=add_n,
    top_k=None,
)
cumsum = _cumsum


# ---------------------------------------------------------------------------------------------------------------
# tf.random / tf.nn
# ---------------------------------------------------------------------------------------------------------------
class _Random(object):
    def __init__(self):
        self._rng = _np.random.default_rng(0)

    def set_seed(self, seed):
        self._rng = _np.random.default_rng(seed)

    def uniform(self, shape, minval=0, maxval=None, dtype=float32, seed=None):    # noqa: A002
        dtype = _np.dtype(dtype)
        if dtype.kind in "iu":
            return _t(self._rng.integers(minval, maxval, size=_shape_list(shape)).astype(dtype))
        maxval = 1.0 if maxval is None else maxval
        return _t(self._rng.uniform(minval, maxval, size=_shape_list(shape)).astype(dtype))

    def normal(self, shape, mean=0.0, stddev=1.0, dtype=float32, seed=None):      # noqa: A002
        return _t((self._rng.standard_normal(_shape_list(shape)) * stddev + mean).astype(dtype))

    def truncated_normal(self, shape, mean=0.0, stddev=1.0, dtype=float32, seed=None):    # noqa: A002
        v = self._rng.standard_normal(_shape_list(shape))
        bad = _np.abs(v) > 2
        while bad.any():
            v[bad] = self._rng.standard_normal(int(bad.sum()))
            bad = _np.abs(v) > 2
        return _t((v * stddev + mean).astype(dtype))

    def shuffle(self, value, seed=None):
        v = _np.array(_a(value))
        self._rng.shuffle(v)
        return _t(v)


random = _Random()


def _relu(x, name=None):
    a = _np.asarray(_a(x))
    return _t(_np.maximum(a, _np.zeros((), dtype=a.dtype)))


def _leaky_relu(x, alpha=0.2, name=None):
    a = _np.asarray(_a(x))
    return _t(_np.where(a > 0, a, a * _np.asarray(alpha, dtype=a.dtype)))


def _sigmoid(x, name=None):
    a = _np.asarray(_a(x))
    return _t((1 / (1 + _np.exp(-a))).astype(a.dtype))


def _softmax(x, axis=-1, name=None):
    a = _np.asarray(_a(x))
    e = _np.exp(a - a.max(axis=axis, keepdims=True))
    return _t(e / e.sum(axis=axis, keepdims=True))


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    a = _np.asarray(_a(x))
    axis = dim if axis is None else axis
    sq = _np.sum(a * a, axis=axis, keepdims=True)
    return _t(a * (1 / _np.sqrt(_np.maximum(sq, _np.asarray(epsilon, dtype=a.dtype)))).astype(a.dtype))


def _dropout(x, rate=None, noise_shape=None, seed=None, name=None, keep_prob=None):
    a = _np.asarray(_a(x))
    if keep_prob is not None and rate is None:
        rate = 1.0 - keep_prob
    if rate == 0:
        return _t(a)
    keep = random._rng.uniform(size=a.shape) >= rate
    return _t(_np.where(keep, a / _np.asarray(1.0 - rate, dtype=a.dtype), _np.zeros((), a.dtype)).astype(a.dtype))


def _top_k(input, k=1, sorted=True):       # noqa: A002
    a = _np.asarray(_a(input))
    idx = _np.argsort(-a, axis=-1, kind="stable")[..., :k]
    return _t(_np.take_along_axis(a, idx, axis=-1)), _t(idx.astype(_np.int32))


math.top_k = _top_k
nn = _Namespace(relu=_relu, leaky_relu=_leaky_relu, sigmoid=_sigmoid, softmax=_softmax, tanh=tanh,
                l2_normalize=_l2_normalize, dropout=_dropout, top_k=_top_k,
                elu=lambda x: _t(_np.where(_a(x) > 0, _a(x), _np.expm1(_a(x)))))
compat = _Namespace(v2=_Namespace(nn=_Namespace(dropout=_dropout)), v1=_Namespace())


# ---------------------------------------------------------------------------------------------------------------
# tf.sparse
# ---------------------------------------------------------------------------------------------------------------
class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices = _t(_np.asarray(_a(indices)).astype(_np.int64).reshape(-1, len(_shape_list(dense_shape))))
        self.values = _t(_a(values))
        self.dense_shape = _t(_np.asarray(_shape_list(dense_shape), dtype=_np.int64))

    @property
    def shape(self):
        return TensorShape(int(s) for s in self.dense_shape)

    @property
    def dtype(self):
        return self.values.dtype

    def get_shape(self):
        return self.shape


def _sp_reorder(sp):
    idx = _np.asarray(sp.indices)
    order = _np.lexsort(tuple(idx[:, k] for k in reversed(_np.arange(idx.shape[1]))))
    return SparseTensor(idx[order], _np.asarray(sp.values)[order], sp.dense_shape)


def _sp_to_dense(sp, default_value=None, validate_indices=True):
    out = _np.zeros(_shape_list(sp.dense_shape), dtype=sp.values.dtype)
    if default_value is not None:
        out[...] = default_value
    idx = _np.asarray(sp.indices)
    out[tuple(idx.T)] = _np.asarray(sp.values)
    return _t(out)


def _sp_dense_matmul(sp_a, b, adjoint_a=False, adjoint_b=False):
    idx = _np.asarray(sp_a.indices)
    r, c = (idx[:, 1], idx[:, 0]) if adjoint_a else (idx[:, 0], idx[:, 1])
    b = _np.asarray(_a(b))
    if adjoint_b:
        b = b.T
    n_out = int(sp_a.dense_shape[1 if adjoint_a else 0])
    out = _np.zeros((n_out, b.shape[1]), dtype=b.dtype)
    _np.add.at(out, r, _np.asarray(sp_a.values)[:, None] * b[c])
    return _t(out)


def _sp_transpose(sp, perm=None):
    idx = _np.asarray(sp.indices)[:, ::-1]
    return _sp_reorder(SparseTensor(idx, sp.values, _shape_list(sp.dense_shape)[::-1]))


def _sp_reduce_sum(sp, axis=None, keepdims=False):
    return reduce_sum(_sp_to_dense_sum(sp), axis=axis, keepdims=keepdims)


def _sp_to_dense_sum(sp):
    out = _np.zeros(_shape_list(sp.dense_shape), dtype=sp.values.dtype)
    _np.add.at(out, tuple(_np.asarray(sp.indices).T), _np.asarray(sp.values))
    return _t(out)


def _sp_concat(axis, sp_inputs):
    nd = len(_shape_list(sp_inputs[0].dense_shape))
    axis = axis % nd
    off, idxs, vals = 0, [], []
    shape = _shape_list(sp_inputs[0].dense_shape)       # noqa: A001
    for sp in sp_inputs:
        i = _np.array(sp.indices)
        i[:, axis] += off
        off += int(sp.dense_shape[axis])
        idxs.append(i)
        vals.append(_np.asarray(sp.values))
    shape[axis] = off
    return _sp_reorder(SparseTensor(_np.concatenate(idxs, 0), _np.concatenate(vals, 0), shape))


def _sp_slice(sp, start, size):     # noqa: A002
    idx = _np.asarray(sp.indices)
    start, size = _np.asarray(_shape_list(start)), _np.asarray(_shape_list(size))
    keep = _np.all((idx >= start) & (idx < start + size), axis=1)
    shape = _np.minimum(size, _np.asarray(_shape_list(sp.dense_shape)) - start)    # noqa: A001
    return SparseTensor(idx[keep] - start, _np.asarray(sp.values)[keep], shape)


sparse = _Namespace(SparseTensor=SparseTensor, reorder=_sp_reorder, to_dense=_sp_to_dense,
                    sparse_dense_matmul=_sp_dense_matmul, transpose=_sp_transpose, reduce_sum=_sp_reduce_sum,
                    concat=_sp_concat, slice=_sp_slice)


# ---------------------------------------------------------------------------------------------------------------
# tf.lookup, tf.errors
# ---------------------------------------------------------------------------------------------------------------
class _KeyValueTensorInitializer(object):
    def __init__(self, keys, values, key_dtype=None, value_dtype=None):
        self.keys, self.values = _np.asarray(_a(keys)), _np.asarray(_a(values))


class _StaticHashTable(object):
    def __init__(self, initializer, default_value):
        self._d = dict(zip(initializer.keys.tolist(), initializer.values.tolist()))
        self._default = default_value
        self._vdtype = initializer.values.dtype

    def lookup(self, keys):
        k = _np.asarray(_a(keys))
        out = _np.array([self._d.get(v, self._default) for v in k.reshape(-1).tolist()], dtype=self._vdtype)
        return _t(out.reshape(k.shape))

    __getitem__ = lookup


lookup = _Namespace(StaticHashTable=_StaticHashTable, KeyValueTensorInitializer=_KeyValueTensorInitializer)


class _Errors(object):
    class InvalidArgumentError(Exception):
        pass

    class OpError(Exception):
        pass


errors = _Errors()


class GradientTape(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("the numpy stand-in for tensorflow has no autodiff; forward semantics only")


# ---------------------------------------------------------------------------------------------------------------
# tf.keras (just enough for tfg.layers.*: weight creation in build(), call dispatch)
# ---------------------------------------------------------------------------------------------------------------
def _glorot_uniform(shape, dtype):      # noqa: A002
    fan_in, fan_out = (shape[0], shape[-1]) if len(shape) >= 2 else (shape[0], shape[0])
    limit = _np.sqrt(6.0 / (fan_in + fan_out))
    return random._rng.uniform(-limit, limit, size=shape).astype(dtype)


_INITIALIZERS = {
    "glorot_uniform": _glorot_uniform,
    "zeros": lambda shape, dtype: _np.zeros(shape, dtype),      # noqa: A002
    "ones": lambda shape, dtype: _np.ones(shape, dtype),        # noqa: A002
}


class _Layer(object):
    def __init__(self, *args, **kwargs):
        unknown = set(kwargs) - {"name", "dtype", "trainable", "dynamic"}
        if unknown:
            # tf.keras.Model.__init__ rejects unknown kwargs with a TypeError (e.g. the stale drop_rate= call sites)
            raise TypeError("('Keyword argument not understood:', {!r})".format(sorted(unknown)[0]))
        object.__setattr__(self, "_stub_weights", [])
        object.__setattr__(self, "_stub_sublayers", [])
        self.built = False
        self.name = kwargs.get("name", self.__class__.__name__.lower())

    def __setattr__(self, key, value):
        if isinstance(value, _Layer) and hasattr(self, "_stub_sublayers"):
            self._stub_sublayers.append(value)
        object.__setattr__(self, key, value)

    def add_weight(self, name=None, shape=None, dtype=None, initializer=None, regularizer=None,      # noqa: A002
                   trainable=True, **kwargs):
        dtype = float32 if dtype is None else dtype
        shape = [int(s) for s in shape]     # noqa: A001
        if initializer is None:
            initializer = "glorot_uniform"
        init = _INITIALIZERS[initializer] if isinstance(initializer, str) else initializer
        v = Variable(init(shape, dtype), trainable=trainable, name=name)
        self._stub_weights.append(v)
        return v

    def build(self, input_shapes):
        return None

    def call(self, inputs, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            def shp(v):
                if v is None:
                    return None
                if isinstance(v, (list, tuple)):
                    return [shp(u) for u in v]
                if hasattr(v, "shape"):
                    return TensorShape(int(s) for s in v.shape)
                return TensorShape(_np.shape(v))
            self.build(shp(inputs))
            self.built = True
        return self.call(inputs, *args, **kwargs)

    @property
    def trainable_variables(self):
        out = [w for w in self._stub_weights if w.trainable]
        for s in self._stub_sublayers:
            out += s.trainable_variables
        return out

    variables = weights = trainable_weights = trainable_variables


class _Dense(_Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_regularizer=None, bias_regularizer=None,
                 **kwargs):
        super().__init__(**kwargs)
        self.units, self.activation, self.use_bias = units, _activation(activation), use_bias

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", [input_shape[-1], self.units])
        if self.use_bias:
            self.bias = self.add_weight("bias", [self.units], initializer="zeros")

    def call(self, inputs, training=None):
        h = matmul(inputs, self.kernel)
        if self.use_bias:
            h = h + self.bias
        return h if self.activation is None else self.activation(h)


class _Dropout(_Layer):
    def __init__(self, rate, **kwargs):
        super().__init__(**kwargs)
        self.rate = rate

    def call(self, inputs, training=None):
        return _dropout(inputs, self.rate) if training else _t(inputs)


class _Sequential(_Layer):
    def __init__(self, layers=None, **kwargs):
        super().__init__(**kwargs)
        self.layers = list(layers or [])
        self._stub_sublayers.extend(self.layers)

    def add(self, layer):
        self.layers.append(layer)
        self._stub_sublayers.append(layer)

    def call(self, inputs, training=None):
        h = inputs
        for layer in self.layers:
            h = layer(h, training=training)
        return h


def _activation(a):
    if a is None or callable(a):
        return a
    return {"relu": _relu, "sigmoid": _sigmoid, "tanh": tanh, "softmax": _softmax, "linear": None}[a]


class _Unavailable(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("not provided by the numpy stand-in for tensorflow")


keras = _Namespace(
    Model=_Layer,
    Sequential=_Sequential,
    layers=_Namespace(Layer=_Layer, Dense=_Dense, Dropout=_Dropout, LSTM=_Unavailable,
                      Activation=lambda a: _activation(a)),
    activations=_Namespace(relu=_relu, sigmoid=_sigmoid, tanh=tanh, softmax=_softmax),
    utils=_Namespace(get_file=_Unavailable),
    regularizers=_Namespace(l2=lambda l=0.01: None, l1=lambda l=0.01: None),
    initializers=_Namespace(),
    optimizers=_Namespace(Adam=_Unavailable),
)
