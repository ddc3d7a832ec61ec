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


# 08 = BASELINE config 5 (19x19, 1600 sims, 32-leaf rounds); 09..11 are the two-model games; 12..14: 13x13 / 7x7, energies 4 / 16
ASYNC_FILES = ["async_%02d.npz" % i for i in list(range(9)) + [12, 13, 14]]


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


# ---- the reference's unit tests recorded as data (tests/golden/units_S9.npz, gen_golden.py child_units) ----------
_TYPES = {0: int, 1: float, 2: np.float32, 3: np.float64, 4: int}


def deser_tree(ints, flts):
    """Rebuilds a dict tree from gen_golden.ser_tree rows, with the recorded key set and scalar types per node."""
    root = None
    stack = []
    for (depth, action, m, index, count, vloss, has_parent, tv, tm, tp), (value, mean, p) in zip(ints, flts):
        node = {}
        if m & 1:
            node['index'] = int(index)
        if m & 2:
            node['count'] = int(count)
        if m & 4:
            node['virtual_loss'] = int(vloss)
        if m & 8:
            node['value'] = _TYPES[int(tv)](value)
        if m & 16:
            node['mean_value'] = _TYPES[int(tm)](mean)
        if m & 32:
            node['p'] = _TYPES[int(tp)](p)
        node['subtree'] = None if m & 128 else {}
        del stack[int(depth):]
        if stack:
            stack[-1]['subtree'][int(action)] = node
            if m & 64:
                node['parent'] = stack[-1] if has_parent else None
        else:
            root = node
            if m & 64:
                node['parent'] = None
        stack.append(node)
    return root


def ser_tree(root):
    """Same rows as tests/golden/gen_golden.py:ser_tree (kept in step with it)."""
    def tcode(v):
        if isinstance(v, np.float32):
            return 2
        if isinstance(v, np.float64):
            return 3
        if isinstance(v, np.integer):
            return 4
        if isinstance(v, float):
            return 1
        return 0
    ints, flts = [], []

    def row(node, depth, action):
        m = 0
        for bit, k in ((1, "index"), (2, "count"), (4, "virtual_loss"), (8, "value"), (16, "mean_value"), (32, "p"), (64, "parent")):
            if k in node:
                m |= bit
        if node.get("subtree") is None:
            m |= 128
        ints.append([depth, action, m, node.get("index", 0), node.get("count", 0), node.get("virtual_loss", 0),
                     1 if node.get("parent") is not None else 0,
                     tcode(node.get("value", 0)), tcode(node.get("mean_value", 0)), tcode(node.get("p", 0))])
        flts.append([float(node.get("value", 0)), float(node.get("mean_value", 0)), float(node.get("p", 0))])
        for a, c in (node.get("subtree") or {}).items():
            row(c, depth + 1, int(a))

    row(root, 0, -1)
    return np.array(ints, dtype=np.int64), np.array(flts, dtype=np.float64)


def unit_calls(z):
    """The recorded rule / symmetry calls: dicts {fn, test, ins: [(kind, array)], out: (kind, array)}."""
    out = []
    for i in range(int(z["n_calls"])):
        p = "c%03d_" % i
        ins = [(int(z[p + "in%d_k" % j]), z[p + "in%d" % j]) for j in range(int(z[p + "nin"]))]
        out.append({"fn": name_of(z, p + "fn"), "test": name_of(z, p + "test"), "ins": ins,
                    "out": (int(z[p + "out_k"]), z[p + "out"])})
    return out


def unit_value(kv):
    """(kind, array) -> python value: 0 None, 1 ndarray, 2 int, 3 list of tuples, 4 dict, 5 float"""
    kind, arr = kv
    if kind == 0:
        return None
    if kind == 1:
        return np.array(arr)
    if kind == 2:
        return int(arr)
    if kind == 3:
        return [tuple(int(t) for t in e) for e in arr]
    if kind == 4:
        return {int(k): int(v) for k, v in arr}
    return float(arr)


def read_sample(path):
    """(board, policy_target, value_target) of a sample.h5, through whatever HDF5 reader exists: h5py, the HDF5 C library
    (h5lite), or -- for files of the pure-Python writer only -- hdf5_min's own reader."""
    try:
        import h5py as mod
    except Exception:
        from sejonggo_amd import h5lite as mod
        if not mod.available():
            from sejonggo_amd.hdf5_min import read_datasets
            r = read_datasets(path)
            return r['board'], r['policy_target'], r['value_target']
    with mod.File(path, "r") as f:
        return f['board'][...], f['policy_target'][...], f['value_target'][()]
