"""Drop-in mirror of the reference's symmetry.py on MI355X (libsgo_hip.so).

k: 0 _id, 1 left_diagonal, 2 vertical_axis, 3 horizontal_axis, 4 rotation_90, 5 rotation_180,
6 rotation_270 -- the order of symmetry.SYMMETRIES (symmetry.py:117-125) -- and 7 right_diagonal, which
the reference implements and tests but does not list.  `vertical_axis` / `horizontal_axis` /
`reverse_*` mutate their argument in place like the reference (symmetry.py:54-56,77-79,50-51).
"""
import ctypes as C
from random import choice

import numpy as np

from . import _lib

N_SYMMETRIES = 7  # what random_symmetry_predict draws from


def sym_lut(size, k):
    lib = _lib.load()
    lut = np.zeros(size * size + 1, dtype=np.int32)
    _lib.check(lib.sgo_sym_lut(C.c_int(size), C.c_int(k), _lib.ptr(lut)), "sgo_sym_lut")
    return lut


def _apply(k, board):
    lib = _lib.require_gpu()
    b = np.ascontiguousarray(board, dtype=np.int32)
    S = b.shape[-2]
    out = np.empty_like(b)
    _lib.check(lib.sgo_sym_apply(C.c_int(S), C.c_int(k), C.c_int(b.shape[0]), _lib.ptr(b), _lib.ptr(out)), "sgo_sym_apply")
    return out.astype(board.dtype, copy=False)


def _invert(k, policy, size=None):
    lib = _lib.require_gpu()
    p = np.ascontiguousarray(policy, dtype=np.float32)
    A = p.shape[-1]
    S = size or int(round((A - 1) ** 0.5))
    out = np.empty_like(p)
    _lib.check(lib.sgo_sym_invert_policy(C.c_int(S), C.c_int(k), C.c_int(p.shape[0]), _lib.ptr(p), _lib.ptr(out)),
               "sgo_sym_invert_policy")
    policy[:, :] = out
    return policy


def _id(tensor):
    return tensor


def left_diagonal(board):
    return _apply(1, board)


def reverse_left_diagonal(policy):
    return _invert(1, policy)


def vertical_axis(board):
    board[...] = _apply(2, board)
    return board


def reverse_vertical_axis(policy):
    return _invert(2, policy)


def horizontal_axis(board):
    board[...] = _apply(3, board)
    return board


def reverse_horizontal_axis(policy):
    return _invert(3, policy)


def rotation_90(board):
    return _apply(4, board)


def reverse_rotation_90(policy):
    return _invert(4, policy)


def rotation_180(board):
    return _apply(5, board)


def reverse_rotation_180(policy):
    return _invert(5, policy)


def rotation_270(board):
    return _apply(6, board)


def reverse_rotation_270(policy):
    return _invert(6, policy)


def right_diagonal(board):
    return _apply(7, board)


def reverse_right_diagonal(policy):
    return _invert(7, policy)


SYMMETRIES = [
    (_id, _id),
    (left_diagonal, reverse_left_diagonal),
    (vertical_axis, reverse_vertical_axis),
    (horizontal_axis, reverse_horizontal_axis),
    (rotation_90, reverse_rotation_90),
    (rotation_180, reverse_rotation_180),
    (rotation_270, reverse_rotation_270),
]


def random_symmetry_predict(model, board):
    """symmetry.py:127-132: ONE random symmetry for the whole batch, policy mapped back."""
    symmetry, reverse_symmetry = choice(SYMMETRIES)
    symm_board = symmetry(board)
    symm_policy, value = model.predict_on_batch(symm_board)
    policy = reverse_symmetry(np.array(symm_policy, dtype=np.float32))
    return policy, value


# The reference's module-level SWAP tables and their builders (symmetry.py:12-42, :44-114).  The tables come from the library
# (sgo_sym_lut restates the float rotation + round construction); the builders restate the formula for callers that ask
# for other angles.
_SWAP_NAMES = {"LEFT_DIAGONAL_SWAP": 1, "VERTICAL_AXIS_SWAP": 2, "HORIZONTAL_AXIS_SWAP": 3, "ROTATION_90_SWAP": 4,
               "ROTATION_180_SWAP": 5, "ROTATION_270_SWAP": 6, "RIGHT_DIAGONAL_SWAP": 7}


def __getattr__(name):
    if name in _SWAP_NAMES:
        from .conf import conf
        return [int(v) for v in sym_lut(conf['SIZE'], _SWAP_NAMES[name])]
    raise AttributeError(name)


def rotation_indexes(angle, size=None):
    from math import cos, sin
    from .conf import conf
    S = size or conf['SIZE']
    c = (S - 1) / 2
    out = [0] * (S * S + 1)
    for y in range(S):
        for x in range(S):
            nx = cos(angle) * (x - c) - sin(angle) * (y - c) + c
            ny = sin(angle) * (x - c) + cos(angle) * (y - c) + c
            out[x + S * y] = int(round(nx + S * ny))
    out[S * S] = S * S
    return out


def axis_symmetry_indexes(angle, size=None):
    from math import cos, sin
    from .conf import conf
    S = size or conf['SIZE']
    c = (S - 1) / 2
    out = [0] * (S * S + 1)
    for y in range(S):
        for x in range(S):
            nx = cos(2 * angle) * (x - c) + sin(2 * angle) * (y - c) + c
            ny = sin(2 * angle) * (x - c) - cos(2 * angle) * (y - c) + c
            out[x + S * y] = int(round(nx + S * ny))
    out[S * S] = S * S
    return out
