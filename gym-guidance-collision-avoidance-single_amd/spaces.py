"""Minimal `gym.spaces` stand-ins (gym is not a dependency).

Only what the reference's callers touch: `shape`, `dtype`, `n`, `low/high`,
`contains`, `sample`.  `Box.contains` keeps the gym<=0.12 semantics the
reference was written against (shape match + inclusive bounds).
If a real gym/gymnasium is importable, its spaces work just as well.
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            self.low = np.asarray(low).astype(dtype)
            self.high = np.asarray(high).astype(dtype)
            self.shape = self.low.shape
        else:
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=dtype)
            self.high = np.full(self.shape, high, dtype=dtype)
        self.dtype = np.dtype(dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all()) and bool((x <= self.high).all())

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        try:
            xi = int(x)
        except (TypeError, ValueError):
            return False
        return 0 <= xi < self.n

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return "Discrete(%d)" % self.n
