"""Deterministic stand-in policy/value nets (the reference's tests use the same idea:
``DummyModel`` in test/tests.py:34-49).

Every net here honours the model contract of the reference (model.py:57,80,90):
``predict_on_batch(X[n,S,S,17]) -> [policy float32[n,S*S+1], value float32[n,1]]`` and a
``.name``.  They exist so that self-play can be driven without trained weights and so that the
GPU engine, the CPU oracle and the Python reference can be fed *bit-identical* priors:

* ``UniformNet``   p = 1/(S*S+1), v = 0             (BASELINE.json configs[0] "uniform-random stub")
* ``DummyNet``     p ∝ reversed(range(1, A+1)), v=1 (restates DummyModel, tests.py:34-49)
* ``HashNet``      p, v are dyadic rationals computed from an integer hash of the 17 planes, so
                   no floating-point rounding happens anywhere and numpy / torch-CPU /
                   torch-ROCm agree to the bit.

Each class works on numpy arrays and on torch tensors (any device); the arithmetic is integer
until the final exact scaling.
"""
import numpy as np

_HASH_MOD = 1000003


def _hash_weights(size):
    """Fixed integer weight table [S,S,17], values in [1, 2^20)."""
    n = size * size * 17
    w = np.empty(n, dtype=np.int64)
    s = 88172645463325252 & 0xFFFFFFFFFFFFFFFF
    for i in range(n):  # xorshift64, seed fixed
        s ^= (s << 13) & 0xFFFFFFFFFFFFFFFF
        s ^= s >> 7
        s ^= (s << 17) & 0xFFFFFFFFFFFFFFFF
        w[i] = (s >> 11) % ((1 << 20) - 1) + 1
    return w.reshape(size, size, 17)


class _Base(object):
    def __init__(self, size, name):
        self.size = size
        self.A = size * size + 1
        self.name = name

    def predict(self, X):
        p, v = self.predict_on_batch(X)
        return p[0], v[0]

    def _on(self, device):
        """The constant prior vector on `device`, copied there once (a per-call host-to-device copy would also be illegal inside
        a captured launch chain)."""
        import torch
        cache = self.__dict__.setdefault("_dev", {})
        if device not in cache:
            cache[device] = torch.from_numpy(self._p).to(device)
        return cache[device]


class UniformNet(_Base):
    def __init__(self, size, name="uniform_stub"):
        _Base.__init__(self, size, name)
        self._p = np.full((self.A,), np.float32(1.0) / np.float32(self.A), dtype=np.float32)

    def predict_on_batch(self, X):
        n = X.shape[0]
        if isinstance(X, np.ndarray):
            return np.tile(self._p, (n, 1)), np.zeros((n, 1), dtype=np.float32)
        import torch
        p = self._on(X.device)
        return p.unsqueeze(0).repeat(n, 1), torch.zeros((n, 1), dtype=torch.float32, device=X.device)


class DummyNet(_Base):
    def __init__(self, size, name="dummy_model"):
        _Base.__init__(self, size, name)
        pol = np.zeros((1, self.A), dtype=np.float32)
        pol[0, :] = list(reversed(range(1, self.A + 1)))
        pol[:, :] /= np.sum(pol, axis=1)[:, np.newaxis]
        self._p = pol[0].copy()

    def predict_on_batch(self, X):
        n = X.shape[0]
        if isinstance(X, np.ndarray):
            return np.tile(self._p, (n, 1)), np.ones((n, 1), dtype=np.float32)
        import torch
        p = self._on(X.device)
        return p.unsqueeze(0).repeat(n, 1), torch.ones((n, 1), dtype=torch.float32, device=X.device)


class HashNet(_Base):
    """Board-dependent, rounding-free pseudo net."""

    def __init__(self, size, name="hash_stub", variant=0):
        _Base.__init__(self, size, name)
        self._w = _hash_weights(size)
        if variant:                           # a second, different pseudo net (two-model evaluation games)
            self._w = np.roll(self._w.reshape(-1), 7 * variant).reshape(self._w.shape) + variant
        self._mul = (np.arange(self.A, dtype=np.int64) * (7919 + 2 * variant) + 13 + variant)
        self._wt = {}

    def predict_on_batch(self, X):
        if isinstance(X, np.ndarray):
            x = X.astype(np.int64).reshape(X.shape[0], -1)
            h = x @ self._w.reshape(-1)
            base = np.mod(h, _HASH_MOD)
            raw = np.mod((base[:, None] + 1) * self._mul[None, :], 1009) + 1
            p = raw.astype(np.float32) * np.float32(2.0 ** -19)
            v = (np.mod(base, 2049) - 1024).astype(np.float32) * np.float32(2.0 ** -10)
            return p, v.reshape(-1, 1)
        import torch
        dev = X.device
        if dev not in self._wt:
            self._wt[dev] = (torch.from_numpy(self._w.reshape(-1)).to(dev),
                             torch.from_numpy(self._mul).to(dev))
        w, mul = self._wt[dev]
        x = X.to(torch.int64).reshape(X.shape[0], -1)
        h = (x * w.unsqueeze(0)).sum(dim=1)
        base = torch.remainder(h, _HASH_MOD)
        raw = torch.remainder((base.unsqueeze(1) + 1) * mul.unsqueeze(0), 1009) + 1
        p = raw.to(torch.float32) * (2.0 ** -19)
        v = (torch.remainder(base, 2049) - 1024).to(torch.float32) * (2.0 ** -10)
        return p, v.reshape(-1, 1)


def make_stub(kind, size):
    if kind == "hash2":
        return HashNet(size, name="hash_stub_2", variant=1)
    return {"uniform": UniformNet, "dummy": DummyNet, "hash": HashNet}[kind](size)
