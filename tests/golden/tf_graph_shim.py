"""A graph-mode ``tensorflow`` stand-in on torch-CPU autograd -- TEST INFRASTRUCTURE, build container only.

TensorFlow 1.12 (the reference's pinned dependency, requirements.txt:4) has no wheel for this Python
and there is no network, so the reference's model files cannot run on the real thing.  This module is
registered as ``sys.modules['tensorflow']`` by ``make_golden_graph.py`` so that the reference's OWN
``autoencoder/autoencoder.py``, ``autoencoder_triplet.py``, ``triplet_loss_utils.py`` and ``utils.py``
execute UNCHANGED: ``_build_model`` builds a lazy graph of ``Node`` objects, ``tf.Session().run(fetches,
feed_dict)`` evaluates it with torch tensors, ``Optimizer.minimize(cost)`` differentiates the cost the
reference built with ``torch.autograd`` and applies the update rule.  Only the op kernels underneath the
reference's code are torch instead of Eigen; the graph (encode / decode / miners / weighted loss / cost /
tied-weight gradients), the epoch loop, the RNG order and the feeds are the reference's.

What is NOT the reference here (and is therefore documented as "restated from TF 1.12 semantics"):
  * op gradients follow TF's registered gradients where they differ from torch defaults:
    ``tf.maximum`` routes the gradient to x where x >= y (MaximumGrad), ``reduce_max/min`` split the
    gradient equally among ties (``torch.amax/amin`` do the same);
  * the four ``tf.train`` optimizers' update rules (GradientDescent, Adagrad with
    initial_accumulator_value=0.1, Momentum without Nesterov, Adam with beta .9/.999, eps 1e-8);
  * ``tf.random_uniform`` (Xavier init) cannot be reproduced: variables are INJECTED by name through
    ``INJECT`` ({'enc-w': W0, 'hidden-bias': bh0, 'visible-bias': bv0}).
``set_dtype(torch.float64)`` evaluates the same graph in double precision (the "truth" vectors);
``torch.float32`` mimics TF's arithmetic width.
"""
import contextlib
import types

import numpy as np
import torch

DT = torch.float32
INJECT = {}          # variable name -> initial value (np array) overriding the graph's initializer
VARIABLES = []       # every tf.Variable created since reset()
_DEFAULT_SESSION = []


def set_dtype(dt):
    global DT
    DT = dt


def reset():
    """Forget all variables (call before building a new model)."""
    VARIABLES.clear()
    INJECT.clear()


class _StaticShape:
    def __init__(self, known):
        self.known = dict(known)

    def __getitem__(self, i):
        return self.known.get(i, None)


class Node:
    """A symbolic tensor: fn(*evaluated inputs) -> torch tensor (or python object)."""

    def __init__(self, fn, inputs=(), name=None, static=None):
        self.fn, self.inputs, self.name, self._static = fn, tuple(inputs), name, static or {}

    # ---- evaluation ----
    def _eval(self, ctx):
        k = id(self)
        if k in ctx:
            return ctx[k]
        args = [_ev(i, ctx) for i in self.inputs]
        v = self.fn(*args)
        ctx[k] = v
        return v

    def eval(self, feed_dict=None, session=None):
        s = session or (_DEFAULT_SESSION[-1] if _DEFAULT_SESSION else Session())
        return s.run(self, feed_dict=feed_dict)

    @property
    def shape(self):
        return _StaticShape(self._static)

    # ---- operators (python scalars take the tensor's dtype, like TF's constant conversion) ----
    def __add__(self, o): return Node(lambda a, b: a + b, (self, o))
    def __radd__(self, o): return Node(lambda a, b: b + a, (self, o))
    def __sub__(self, o): return Node(lambda a, b: a - b, (self, o))
    def __rsub__(self, o): return Node(lambda a, b: b - a, (self, o))
    def __mul__(self, o): return Node(lambda a, b: a * b, (self, o))
    def __rmul__(self, o): return Node(lambda a, b: b * a, (self, o))
    def __truediv__(self, o): return Node(lambda a, b: a / b, (self, o))
    def __rtruediv__(self, o): return Node(lambda a, b: b / a, (self, o))
    def __neg__(self): return Node(lambda a: -a, (self,))
    def __getitem__(self, idx): return Node(lambda a: a[idx], (self,))
    __hash__ = object.__hash__


def _ev(x, ctx):
    return x._eval(ctx) if isinstance(x, Node) else x


def _node(fn, *inputs, **kw):
    return Node(fn, inputs, **kw)


class _Placeholder(Node):
    def __init__(self, name, is_sparse):
        super().__init__(None, (), name=name)
        self.is_sparse = is_sparse

    def _eval(self, ctx):
        if id(self) not in ctx:
            raise RuntimeError("placeholder %r was not fed" % (self.name,))
        return ctx[id(self)]


def _feed_value(ph, v):
    if ph.is_sparse:
        if isinstance(v, tuple):                               # (indices [nnz,2], values, shape): utils.get_sparse_ind_val_shape
            ind, val, shp = v
            i = torch.as_tensor(np.asarray(ind, np.int64).T.copy())
            return torch.sparse_coo_tensor(i, torch.as_tensor(np.asarray(val, np.float64)).to(DT), tuple(int(s) for s in shp)).coalesce()
        raise TypeError("sparse placeholder fed with %r" % (type(v),))
    if hasattr(v, "to_numpy"):
        v = v.to_numpy()
    return torch.as_tensor(np.asarray(v, np.float64)).to(DT)   # tf.placeholder('float'): cast to float32


class _Variable(Node):
    def __init__(self, init, name):
        super().__init__(None, (), name=name)
        self.init = init
        self.tensor = None
        VARIABLES.append(self)

    def initialize(self):
        if self.name in INJECT:
            v = torch.as_tensor(np.asarray(INJECT[self.name], np.float64)).to(DT)
        else:
            v = _ev(self.init, {})
            v = torch.as_tensor(np.asarray(v, np.float64)).to(DT) if not torch.is_tensor(v) else v.to(DT)
        self.tensor = v.clone().requires_grad_(True)

    def _eval(self, ctx):
        if self.tensor is None:
            raise RuntimeError("variable %r used before tf.global_variables_initializer ran" % (self.name,))
        return self.tensor

    def numpy(self):
        return self.tensor.detach().cpu().numpy().copy()


class _TFMaximum(torch.autograd.Function):
    """tf.maximum with TF's MaximumGrad: dx = g * (x >= y), dy = g * (x < y)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return torch.maximum(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        m = (x >= y).to(g.dtype)
        gx, gy = g * m, g * (1 - m)
        while gx.dim() > x.dim():
            gx = gx.sum(0)
        while gy.dim() > y.dim():
            gy = gy.sum(0)
        if gx.shape != x.shape:
            gx = gx.sum_to_size(x.shape)
        if gy.shape != y.shape:
            gy = gy.sum_to_size(y.shape)
        return gx, gy


def _t(x, like=None):
    """python scalar -> 0-d tensor of the working dtype."""
    if torch.is_tensor(x):
        return x
    return torch.tensor(x, dtype=(like.dtype if like is not None and like.is_floating_point() else DT))


def _axis(a):
    return tuple(a) if isinstance(a, (list, tuple)) else a


def _reduce(fn_all, fn_axis):
    def op(x, axis=None, keepdims=False):
        def run(v):
            if axis is None:
                return fn_all(v)
            return fn_axis(v, _axis(axis), keepdims)
        return _node(run, x)
    return op


class _TrainOp(Node):
    def __init__(self, optimizer, cost):
        super().__init__(None, (), name="train_step")
        self.optimizer, self.cost = optimizer, cost


class _Optimizer:
    def __init__(self, learning_rate):
        self.lr = float(learning_rate)
        self.slots = {}
        self.t = 0

    def minimize(self, cost):
        return _TrainOp(self, cost)

    def apply(self, cost_value):
        params = [v.tensor for v in VARIABLES]
        grads = torch.autograd.grad(cost_value, params, retain_graph=True, allow_unused=True)
        self.t += 1
        with torch.no_grad():
            for v, g in zip(VARIABLES, grads):
                if g is None:
                    continue
                self.update(v, g)
        return [None if g is None else g.detach().clone() for g in grads]


class GradientDescentOptimizer(_Optimizer):
    def update(self, v, g):
        v.tensor -= self.lr * g


class AdagradOptimizer(_Optimizer):
    def __init__(self, learning_rate, initial_accumulator_value=0.1):
        super().__init__(learning_rate)
        self.init_acc = initial_accumulator_value

    def update(self, v, g):
        acc = self.slots.setdefault(id(v), torch.full_like(v.tensor, self.init_acc))
        acc += g * g
        v.tensor -= self.lr * g * torch.rsqrt(acc)


class MomentumOptimizer(_Optimizer):
    def __init__(self, learning_rate, momentum):
        super().__init__(learning_rate)
        self.momentum = float(momentum)

    def update(self, v, g):
        acc = self.slots.setdefault(id(v), torch.zeros_like(v.tensor))
        acc *= self.momentum
        acc += g
        v.tensor -= self.lr * acc


class AdamOptimizer(_Optimizer):
    def update(self, v, g):
        b1, b2, eps = 0.9, 0.999, 1e-8
        m, s = self.slots.setdefault(id(v), (torch.zeros_like(v.tensor), torch.zeros_like(v.tensor)))
        lr_t = self.lr * np.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        m *= b1; m += (1 - b1) * g
        s *= b2; s += (1 - b2) * g * g
        v.tensor -= lr_t * m / (torch.sqrt(s) + eps)


class Session:
    graph = None

    def __enter__(self):
        _DEFAULT_SESSION.append(self)
        return self

    def __exit__(self, *a):
        _DEFAULT_SESSION.pop()
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        ctx = {}
        for ph, v in (feed_dict or {}).items():
            ctx[id(ph)] = _feed_value(ph, v)
        out = [None] * len(fl)
        for k, f in enumerate(fl):                              # forward fetches first: values BEFORE the update
            if isinstance(f, _TrainOp) or f is None:
                continue
            out[k] = _to_numpy(f._eval(ctx))
        for k, f in enumerate(fl):
            if isinstance(f, _TrainOp):
                self.last_grads = f.optimizer.apply(f.cost._eval(ctx))
        return out[0] if single else out


def _to_numpy(v):
    if torch.is_tensor(v):
        if v.is_sparse:
            v = v.to_dense()
        a = v.detach().cpu().numpy()
        return a.copy() if a.ndim else a[()]
    return v


class _Saver:
    def save(self, session, path):
        np.savez(path + ".shim.npz", **{v.name: v.numpy() for v in VARIABLES})
        return path

    def restore(self, session, path):
        z = np.load(path + ".shim.npz")
        for v in VARIABLES:
            v.tensor = torch.as_tensor(z[v.name]).to(DT).clone().requires_grad_(True)


class _Writer:
    def __init__(self, *a, **k): pass
    def add_summary(self, *a, **k): pass
    def close(self): pass


def _gradients(ys, xs):
    xs_l = xs if isinstance(xs, (list, tuple)) else [xs]

    def make(i):
        def run(y, *xv):
            return torch.autograd.grad(y, list(xv), retain_graph=True, allow_unused=True)[i]
        return Node(run, (ys,) + tuple(xs_l))
    return [make(i) for i in range(len(xs_l))]


def _l2_normalize(x, axis, epsilon=1e-12):
    def run(v):
        ss = torch.sum(v * v, dim=axis, keepdim=True)
        return v * torch.rsqrt(_TFMaximum.apply(ss, _t(epsilon, v)))
    return _node(run, x)


def make_module():
    tf = types.ModuleType("tensorflow")
    tf.__shim__ = True
    tf.bool, tf.float32, tf.int32 = torch.bool, torch.float32, torch.int32

    def cast(x, dtype):
        def run(v):
            if dtype is torch.float32 or dtype == "float":
                return v.to(DT)
            return v.to(dtype)
        return _node(run, x)
    tf.cast = cast
    tf.to_float = lambda x: _node(lambda v: (v if torch.is_tensor(v) else torch.as_tensor(v)).to(DT), x)
    tf.eye = lambda n: _node(lambda k: torch.eye(int(k), dtype=DT), n)
    tf.shape = lambda x: _node(lambda v: torch.as_tensor(tuple(v.shape)), x)
    tf.ones = lambda shape: _node(lambda s: torch.ones(tuple(int(i) for i in np.atleast_1d(np.asarray(s))), dtype=DT), shape)
    tf.zeros = lambda shape: _node(lambda s: torch.zeros(tuple(int(i) for i in np.atleast_1d(np.asarray(s))), dtype=DT), shape)
    tf.logical_not = lambda x: _node(torch.logical_not, x)
    tf.logical_and = lambda a, b: _node(torch.logical_and, a, b)
    tf.equal = lambda a, b: _node(lambda u, v: u == v, a, b)
    tf.greater = lambda a, b: _node(lambda u, v: u > _t(v, u), a, b)
    tf.expand_dims = lambda x, axis: Node(lambda v: v.unsqueeze(axis), (x,), static={axis: 1})
    tf.transpose = lambda x: _node(lambda v: v.t(), x)
    tf.matmul = lambda a, b: _node(torch.matmul, a, b)
    tf.multiply = lambda a, b: _node(lambda u, v: u * v, a, b)
    tf.maximum = lambda a, b: _node(lambda u, v: _TFMaximum.apply(u, _t(v, u)), a, b)
    tf.squeeze = lambda x: _node(lambda v: v.squeeze(), x)
    tf.log = lambda x: _node(torch.log, x)
    tf.log_sigmoid = lambda x: _node(torch.nn.functional.logsigmoid, x)
    tf.squared_difference = lambda a, b: _node(lambda u, v: (u - v) * (u - v), a, b)
    tf.reduce_sum = _reduce(torch.sum, lambda v, ax, kd: torch.sum(v, dim=ax, keepdim=kd))
    tf.reduce_mean = _reduce(torch.mean, lambda v, ax, kd: torch.mean(v, dim=ax, keepdim=kd))
    tf.reduce_max = _reduce(torch.amax, lambda v, ax, kd: torch.amax(v, dim=ax, keepdim=kd))     # ties share the gradient, like TF
    tf.reduce_min = _reduce(torch.amin, lambda v, ax, kd: torch.amin(v, dim=ax, keepdim=kd))
    tf.nn = types.SimpleNamespace(sigmoid=lambda x: _node(torch.sigmoid, x), tanh=lambda x: _node(torch.tanh, x),
                                  l2_normalize=_l2_normalize)
    tf.placeholder = lambda dtype, name=None: _Placeholder(name, False)
    tf.sparse = types.SimpleNamespace(
        placeholder=lambda dtype, name=None: _Placeholder(name, True),
        matmul=lambda sp, w: _node(lambda s, v: torch.sparse.mm(s, v), sp, w),
        to_dense=lambda sp: _node(lambda s: s.to_dense(), sp),
        reduce_sum=lambda sp, axis=None: _node(lambda s: torch.sparse.sum(s) if axis is None else torch.sparse.sum(s, dim=axis).to_dense(), sp))
    tf.Variable = lambda init, name=None: _Variable(init, name)
    tf.random_uniform = lambda shape, minval=0.0, maxval=1.0: _node(
        lambda: torch.as_tensor(np.random.RandomState(0).uniform(minval, maxval, shape)).to(DT))   # placeholder draw; tests INJECT W0
    tf.global_variables_initializer = lambda: _node(lambda: [v.initialize() for v in VARIABLES] and None)
    tf.set_random_seed = lambda s: None
    tf.gradients = _gradients
    tf.name_scope = lambda name: contextlib.nullcontext()
    tf.Session = Session
    tf.summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None,
                                       merge_all=lambda: _node(lambda: b""), FileWriter=_Writer)
    tf.train = types.SimpleNamespace(Saver=_Saver, GradientDescentOptimizer=GradientDescentOptimizer, AdagradOptimizer=AdagradOptimizer,
                                     MomentumOptimizer=MomentumOptimizer, AdamOptimizer=AdamOptimizer)
    return tf
