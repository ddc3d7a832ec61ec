"""Shared helpers for the parity tests (fixture loading, canonical tree hashing)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sha8(arr):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(arr).tobytes()).digest()[:8], dtype=np.uint8)


def unpack_mask(bits, A):
    return np.unpackbits(bits)[:A]


def name_of(z, key):
    return bytes(z[key]).decode()


ASYNC_FILES = ["async_%02d.npz" % i for i in range(8)]
