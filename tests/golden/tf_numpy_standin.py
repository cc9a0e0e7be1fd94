"""A NumPy stand-in for the ~40 TensorFlow symbols the reference's LRGNet path touches.

Used ONLY by tests/golden/make_golden.py, in the build container, to execute the reference's
own unmodified Python (``LrgNet.__init__`` in learn_region_grow_util.py, and the whole
test_region_grow.py / test_random_restart.py scripts) without TensorFlow, which is not
installed anywhere in this environment.  It is this repo's code, not the reference's: a lazy
graph of NumPy closures with TF1 session semantics.  ``tf.nn.conv1d`` with a [1,Cin,Cout]
filter / stride 1 / VALID is evaluated as ``x @ W[0]`` in float32.

install() registers the stand-in as ``tensorflow`` (+ ``tensorflow.keras``), a stub
``metric_loss_ops`` (imported but unused by the LRGNet path) and a stub ``h5py`` whose
``File`` serves in-memory datasets.
"""
import sys
import types
import numpy as np

float32 = np.float32
int32 = np.int32
int64 = np.int64
bool_ = np.bool_

_VARIABLES = {}          # name -> Variable node (current "default graph")
RESTORE_WEIGHTS = {}     # name -> ndarray, consumed by Saver.restore
H5_FILES = {}            # filename -> {dataset: ndarray}
H5_WRITTEN = {}          # filename -> {dataset: ndarray}, what a script wrote through the stub h5py


class Node:
    def __init__(self, fn, inputs=(), name=None):
        self.fn, self.inputs, self.name = fn, tuple(inputs), name

    def eval(self, feed, cache):
        k = id(self)
        if k in cache:
            return cache[k]
        if self in feed:
            v = np.asarray(feed[self])
        else:
            v = self.fn(*[_ev(i, feed, cache) for i in self.inputs])
        cache[k] = v
        return v

    __hash__ = object.__hash__

    def __add__(self, o): return Node(lambda a, b: a + b, (self, o))
    def __radd__(self, o): return Node(lambda a, b: b + a, (self, o))
    def __sub__(self, o): return Node(lambda a, b: a - b, (self, o))
    def __rsub__(self, o): return Node(lambda a, b: b - a, (self, o))
    def __mul__(self, o): return Node(lambda a, b: a * b, (self, o))
    def __rmul__(self, o): return Node(lambda a, b: b * a, (self, o))
    def __truediv__(self, o): return Node(lambda a, b: a / b, (self, o))
    def __gt__(self, o): return Node(lambda a, b: a > b, (self, o))
    def __getitem__(self, idx): return Node(lambda a: a[idx], (self,))


def _ev(x, feed, cache):
    return x.eval(feed, cache) if isinstance(x, Node) else x


class Placeholder(Node):
    def __init__(self, dtype, shape):
        super().__init__(None, ())
        self.dtype, self.shape = dtype, shape

    def eval(self, feed, cache):
        if self not in feed:
            raise KeyError('placeholder not fed')
        v = np.asarray(feed[self], dtype=self.dtype)
        assert tuple(v.shape) == tuple(self.shape), (v.shape, self.shape)
        return v


class Variable(Node):
    def __init__(self, value, name=None):
        super().__init__(None, (), name)
        self.value = np.asarray(value)

    def eval(self, feed, cache):
        return self.value


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _sparse_ce(logits, labels):
    logits = _f32(logits)
    if logits.shape[0] == 0:
        return np.zeros((0,), np.float32)
    m = logits.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(axis=-1))
    picked = np.take_along_axis(logits, np.asarray(labels)[..., None].astype(np.int64), axis=-1)[..., 0]
    return (lse - picked).astype(np.float32)


def _softmax(x, axis=-1):
    x = _f32(x)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _mean(x):
    x = np.asarray(x)
    if x.size == 0:
        return np.float32(np.nan)
    return x.mean(dtype=x.dtype if x.dtype.kind == 'f' else None)


def build_module():
    tf = types.ModuleType('tensorflow')
    tf.float32, tf.int32, tf.int64, tf.bool = float32, int32, int64, bool_

    compat = types.ModuleType('tensorflow.compat')
    v1 = types.ModuleType('tensorflow.compat.v1')
    tf.compat, compat.v1 = compat, v1
    v1.disable_eager_execution = lambda: None
    v1.reset_default_graph = lambda: _VARIABLES.clear()
    v1.placeholder = lambda dtype, shape=None: Placeholder(dtype, shape)

    def get_variable(name, shape, initializer=None, dtype=float32):
        if name in _VARIABLES:
            raise ValueError('Variable %s already exists' % name)
        v = Variable(np.zeros(shape, dtype=dtype), name)
        _VARIABLES[name] = v
        return v
    v1.get_variable = get_variable
    v1.constant_initializer = lambda value=0.0: ('const', value)
    v1.where = lambda cond: Node(lambda c: np.argwhere(c), (cond,))

    class ConfigProto:
        def __init__(self):
            self.gpu_options = types.SimpleNamespace(allow_growth=False)
            self.allow_soft_placement = False
            self.log_device_placement = False
    v1.ConfigProto = ConfigProto

    class Session:
        def __init__(self, config=None):
            pass

        def run(self, fetches, feed_dict=None):
            feed = feed_dict or {}
            cache = {}
            if isinstance(fetches, (list, tuple)):
                return [_ev(f, feed, cache) for f in fetches]
            return _ev(fetches, feed, cache)
    v1.Session = Session

    train = types.ModuleType('tensorflow.compat.v1.train')
    v1.train = train

    class Saver:
        def restore(self, sess, path):
            for name, var in _VARIABLES.items():
                if name in RESTORE_WEIGHTS:
                    w = np.asarray(RESTORE_WEIGHTS[name], dtype=var.value.dtype)
                    assert w.shape == var.value.shape, (name, w.shape, var.value.shape)
                    var.value = w
    train.Saver = Saver

    class AdamOptimizer:
        def __init__(self, lr):
            pass

        def minimize(self, loss, global_step=None):
            return Node(lambda: None, ())
    train.AdamOptimizer = AdamOptimizer

    keras = types.ModuleType('tensorflow.keras')
    keras.initializers = types.SimpleNamespace(VarianceScaling=lambda **kw: ('vs', kw))
    tf.keras = keras

    nn = types.ModuleType('tensorflow.nn')
    tf.nn = nn

    def conv1d(input, filters, stride=1, padding='VALID'):
        assert stride == 1 and padding == 'VALID'

        def run(x, f):
            assert f.ndim == 3 and f.shape[0] == 1 and x.shape[-1] == f.shape[1]
            return np.matmul(_f32(x), _f32(f)[0])
        return Node(run, (input, filters))
    nn.conv1d = conv1d
    nn.bias_add = lambda x, b: Node(lambda a, c: a + c, (x, b))
    nn.relu = lambda x: Node(lambda a: np.maximum(a, 0), (x,))
    nn.softmax = lambda x, axis=-1: Node(lambda a: _softmax(a, axis), (x,))
    nn.sparse_softmax_cross_entropy_with_logits = lambda logits, labels: Node(_sparse_ce, (logits, labels))

    tf.reduce_max = lambda input_tensor, axis=None: Node(lambda a: a.max(axis=axis), (input_tensor,))
    tf.reduce_mean = lambda input_tensor, axis=None: Node(lambda a: _mean(a) if axis is None else a.mean(axis=axis), (input_tensor,))
    tf.reduce_sum = lambda input_tensor, axis=None: Node(lambda a: np.asarray(a).sum(axis=axis), (input_tensor,))

    def concat(axis, values):
        return Node(lambda *vs: np.concatenate(vs, axis=axis), tuple(values))
    tf.concat = concat
    tf.tile = lambda x, multiples: Node(lambda a: np.tile(a, multiples), (x,))
    tf.reshape = lambda x, shape: Node(lambda a: np.reshape(a, shape), (x,))
    tf.cast = lambda x, dtype: Node(lambda a: np.asarray(a).astype(dtype), (x,))

    def gather_nd(params, indices):
        return Node(lambda p, i: p[tuple(np.asarray(i).T)] if len(i) else np.zeros((0,) + p.shape[np.asarray(i).shape[1]:], p.dtype),
                    (params, indices))
    tf.gather_nd = gather_nd

    def cond(pred, true_fn, false_fn):
        t, f = true_fn(), false_fn()
        return Node(lambda p, a, b: a if p else b, (pred, t, f))
    tf.cond = cond
    tf.math = types.SimpleNamespace(is_nan=lambda x: Node(lambda a: bool(np.isnan(a)), (x,)))
    tf.equal = lambda a, b: Node(lambda x, y: np.asarray(x) == np.asarray(y), (a, b))
    tf.argmax = lambda input, axis=None: Node(lambda a: np.argmax(a, axis=axis).astype(np.int64), (input,))
    tf.logical_and = lambda a, b: Node(np.logical_and, (a, b))
    tf.Variable = lambda value, **kw: Variable(value)
    return tf


def install():
    tf = build_module()
    sys.modules['tensorflow'] = tf
    sys.modules['tensorflow.keras'] = tf.keras
    sys.modules['tensorflow.compat'] = tf.compat
    sys.modules['tensorflow.compat.v1'] = tf.compat.v1
    mlo = types.ModuleType('metric_loss_ops')
    mlo.triplet_semihard_loss = lambda *a, **k: None
    sys.modules['metric_loss_ops'] = mlo

    h5 = types.ModuleType('h5py')

    class File:
        def __init__(self, filename, mode='r'):
            self.filename = filename
            if mode == 'w':                      # stage_data.py:251-258 writes its tuples: kept in memory (H5_WRITTEN)
                self.d = H5_WRITTEN[filename] = {}
                return
            if mode != 'r' or filename not in H5_FILES:
                raise IOError('stand-in h5py: %s not registered' % filename)
            self.d = H5_FILES[filename]

        def __getitem__(self, k):
            return self.d[k]

        def create_dataset(self, name, data=None, dtype=None, **kw):
            self.d[name] = np.array(data, dtype=dtype)

        def close(self):
            pass
    h5.File = File
    sys.modules['h5py'] = h5
    return tf
