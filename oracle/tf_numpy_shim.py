"""TEST INFRASTRUCTURE ONLY (build container) -- a NumPy stand-in for the handful of TensorFlow-1 calls that the
reference's policy networks make, so that the REAL function bodies networks.actor_particle (networks.py:517-538) and
networks.actor_checkers / convnet_1 (networks.py:549-578, :67-75) can be executed here and their layer wiring, concat
order, variable names and shapes pinned by golden vectors (oracle/gen_golden_actor.py -> tests/golden/actor_*.npz).

What this does and does not pin: the WIRING is the reference's own code; the SEMANTICS of the primitive ops below are
this file's restatement of the published TF1 behaviour (tf.layers.dense = x @ kernel + bias then activation;
tf.contrib.layers.conv2d = NHWC cross-correlation, "SAME" zero padding, bias, activation; softmax over the last axis).
TensorFlow itself is not installable in the container, so the actor oracles stay "parity unpinned against TensorFlow".
Float32 throughout, like the reference graph's placeholders and variables.
"""
import contextlib
import types

import numpy as np


class T(np.ndarray):
    """ndarray with the two TensorShape calls the reference makes (networks.py:73: conv1.get_shape().as_list())."""

    def get_shape(self):
        shape = self.shape
        return types.SimpleNamespace(as_list=lambda: list(shape))


def _t(x):
    return np.asarray(x, dtype=np.float32).view(T)


class Shim(types.ModuleType):
    """`tf` for one forward pass.  weights: dict var-name -> array; missing variables are created (random, recorded)."""

    def __init__(self, weights=None, rng=None, scale=0.5):
        super().__init__("tensorflow")
        self.weights = {} if weights is None else weights
        self.created = []
        self._rng = rng or np.random.default_rng(0)
        self._scale = scale
        self._scope = []
        self.float32 = np.float32
        self.nn = types.SimpleNamespace(relu=lambda x, name=None: _t(np.maximum(x, np.float32(0))),
                                        softmax=self._softmax, bias_add=lambda x, b: _t(x + b))
        self.layers = types.SimpleNamespace(dense=self._dense)
        self.contrib = types.SimpleNamespace(layers=types.SimpleNamespace(conv2d=self._conv2d))
        self.initializers = types.SimpleNamespace(truncated_normal=lambda *a, **k: None)

    # ---- variables / scopes ----------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name, **kwargs):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def _full(self, name):
        return "/".join(self._scope + [name])

    def get_variable(self, name, shape, dtype=None, initializer=None):
        full = self._full(name)
        if full not in self.weights:
            fan = shape[0] if len(shape) > 1 else 4
            self.weights[full] = (self._rng.standard_normal(tuple(shape)) * self._scale / np.sqrt(fan)).astype(np.float32)
            self.created.append(full)
        w = self.weights[full]
        assert tuple(w.shape) == tuple(shape), (full, w.shape, shape)
        return _t(w)

    # ---- ops ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _softmax(x, name=None):
        z = x - x.max(axis=-1, keepdims=True)
        e = np.exp(z)
        return _t(e / e.sum(axis=-1, keepdims=True))

    def _dense(self, inputs, units, activation=None, use_bias=True, name=None):
        with self.variable_scope(name):
            k = self.get_variable("kernel", [inputs.shape[-1], units])
            y = _t(inputs) @ k
            if use_bias:
                y = y + self.get_variable("bias", [units])
        return _t(activation(y) if activation is not None else y)

    def _conv2d(self, inputs, num_outputs, kernel_size, stride, padding="SAME", activation_fn=None):
        assert padding == "SAME" and list(np.atleast_1d(stride)) in ([1, 1], [1]), "only what networks.convnet_1 uses"
        kh, kw = kernel_size
        x = _t(inputs)
        n, h, w, c = x.shape
        with self.variable_scope("Conv"):                   # tf.contrib.layers.conv2d's default scope name
            k = self.get_variable("weights", [kh, kw, c, num_outputs])
            b = self.get_variable("biases", [num_outputs])
        ph, pw = (kh - 1) // 2, (kw - 1) // 2               # SAME, stride 1, odd kernels: symmetric zero padding
        xp = np.zeros((n, h + kh - 1, w + kw - 1, c), np.float32)
        xp[:, ph:ph + h, pw:pw + w] = x
        out = np.zeros((n, h, w, num_outputs), np.float32)
        for dr in range(kh):
            for dc in range(kw):
                out += xp[:, dr:dr + h, dc:dc + w, :] @ k[dr, dc]
        out = out + b
        return _t(activation_fn(out) if activation_fn is not None else out)

    @staticmethod
    def concat(values, axis):
        return _t(np.concatenate([np.asarray(v, np.float32) for v in values], axis=axis))

    @staticmethod
    def matmul(a, b):
        return _t(np.asarray(a) @ np.asarray(b))

    @staticmethod
    def add_n(xs):
        acc = xs[0]
        for x in xs[1:]:
            acc = acc + x
        return _t(acc)

    @staticmethod
    def reshape(x, shape):
        return _t(np.reshape(x, shape))


def load_networks(shim, ref_root="/root/reference"):
    """Imports the reference's alg/networks.py with `tensorflow` bound to the shim (fresh module object per call)."""
    import importlib.util
    import os
    import sys
    sys.dont_write_bytecode = True
    saved = sys.modules.get("tensorflow")
    sys.modules["tensorflow"] = shim
    try:
        spec = importlib.util.spec_from_file_location("cm3_ref_networks", os.path.join(ref_root, "alg", "networks.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            del sys.modules["tensorflow"]
        else:
            sys.modules["tensorflow"] = saved
    return mod
