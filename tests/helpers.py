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


def dict_tree_hash(root):
    """Canonical serialisation of a host dict tree (same record as tests/golden/gen_golden.py:_tree_hash and
    oracle ora_game_tree_serialize): pre-order, ascending action, <i a, i count, f value, f mean, d p, i vloss, i expanded>."""
    import struct
    h = hashlib.sha1()
    n_nodes = [0]

    def rec(node):
        for a in node["subtree"]:
            c = node["subtree"][a]
            h.update(struct.pack("<iiffdii", int(a), int(c["count"]), float(c["value"]), float(c["mean_value"]),
                                 float(c["p"]), int(c.get("virtual_loss", 0)), 1 if c["subtree"] else 0))
            n_nodes[0] += 1
            if c["subtree"]:
                rec(c)
    rec(root)
    return h.digest()[:16], n_nodes[0]


SYNC_FILES = ["sync_%02d.npz" % i for i in range(3)]
