"""Drop-in mirror of the reference's play.py (rules part) running on MI355X through libsgo_hip.so.

Same names, argument meaning, in-place mutation and error behaviour as the reference:
  game_init (play.py:295-299), make_play (:226-242), legal_moves (:71-104), get_winner (:274-284),
  index2coord / coord2index (:31-37), get_real_board (:106-112).
Boards are caller-owned numpy int32 arrays [1,S,S,17] (batches [n,S,S,17] are accepted too and are
processed in one kernel launch).  SIZE / KOMI are read from sejonggo_amd.conf at call time unless the
board's own shape says otherwise.
"""
import ctypes as C

import numpy as np

from . import _lib
from .conf import conf


def _size_of(board):
    return board.shape[-2]


def index2coord(index, size=None):
    S = size or conf['SIZE']
    y = index // S
    x = index - S * y
    return x, y


def coord2index(x, y, size=None):
    S = size or conf['SIZE']
    return y * S + x


def game_init(size=None):
    lib = _lib.require_gpu()
    S = size or conf['SIZE']
    board = np.zeros((1, S, S, 17), dtype=np.int32)
    _lib.check(lib.sgo_game_init(C.c_int(S), C.c_int(1), _lib.ptr(board)), "sgo_game_init")
    return board, 1


def make_play(x, y, board, color=None):
    """play.py:226-242.  Mutates `board` in place and returns (board, player_who_moved)."""
    lib = _lib.require_gpu()
    if board.dtype != np.int32 or not board.flags['C_CONTIGUOUS']:
        raise TypeError("board must be a C-contiguous int32 array (the reference's game_init dtype)")
    S = _size_of(board)
    n = board.shape[0] if board.ndim == 4 else 1
    xs = np.full(n, x, dtype=np.int32) if np.isscalar(x) else np.ascontiguousarray(x, dtype=np.int32)
    ys = np.full(n, y, dtype=np.int32) if np.isscalar(y) else np.ascontiguousarray(y, dtype=np.int32)
    if color is None:
        cols = np.zeros(n, dtype=np.int32)
    else:
        cols = np.full(n, color, dtype=np.int32) if np.isscalar(color) else np.ascontiguousarray(color, dtype=np.int32)
    movers = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    _lib.check(lib.sgo_make_play(C.c_int(S), C.c_int(n), _lib.ptr(board), _lib.ptr(xs), _lib.ptr(ys), _lib.ptr(cols),
                                 _lib.ptr(movers), _lib.ptr(status)), "sgo_make_play")
    if (status == _lib.SGO_ERR_OCCUPIED).any():
        raise AssertionError("make_play on an occupied point")      # play.py:233-234
    if (status == _lib.SGO_ERR_RANGE).any():
        raise IndexError("make_play outside the board")
    if n == 1:
        return board, int(movers[0])
    return board, movers


def legal_moves(board):
    """play.py:71-104: int64 mask, 1 = illegal, last entry (pass) = 0."""
    lib = _lib.require_gpu()
    b = np.ascontiguousarray(board, dtype=np.int32)
    S = _size_of(b)
    n = b.shape[0] if b.ndim == 4 else 1
    mask = np.zeros((n, S * S + 1), dtype=np.uint8)
    _lib.check(lib.sgo_legal_moves(C.c_int(S), C.c_int(n), _lib.ptr(b), _lib.ptr(mask)), "sgo_legal_moves")
    mask = mask.astype(np.int64)
    return mask[0] if n == 1 else mask


def get_real_board(board):
    player = board[0, 0, 0, -1]
    if player == 1:
        return board[0, :, :, 0] - board[0, :, :, 1]
    return board[0, :, :, 1] - board[0, :, :, 0]


def get_winner(board, komi=None):
    """play.py:274-284 -> (winner, black_points, white_points)."""
    lib = _lib.require_gpu()
    b = np.ascontiguousarray(board, dtype=np.int32)
    S = _size_of(b)
    n = b.shape[0] if b.ndim == 4 else 1
    w = np.zeros(n, dtype=np.int32)
    bl = np.zeros(n, dtype=np.int32)
    wh = np.zeros(n, dtype=np.float64)
    k = conf['KOMI'] if komi is None else komi
    _lib.check(lib.sgo_get_winner(C.c_int(S), C.c_int(n), _lib.ptr(b), C.c_double(k), _lib.ptr(w), _lib.ptr(bl),
                                  _lib.ptr(wh)), "sgo_get_winner")
    if n == 1:
        return int(w[0]), int(bl[0]), float(wh[0])
    return w, bl, wh


# ------------------------------------------------------------------------------------------------------------
# Host-side dict trees (the reference's public tree-node contract, play.py:376-421: nodes are dicts with the keys
# index, count, value, mean_value, p, subtree, parent, virtual_loss).  The production search keeps its trees on
# the GPU (engine.SelfPlayEngine / k_search); these functions serve callers that hold Python dict trees -- the
# reference's sync path (self_play.py), its unit tests and GTP-style front-ends.  The expensive part of building a
# node, the legal-move mask, still runs on the GPU (legal_moves above).  Arithmetic follows the reference's scalar
# expressions so that numpy's scalar promotion gives the same float32 / float64 regime per child.
# ------------------------------------------------------------------------------------------------------------
Cpuct = 1


def _new_node(index, p, parent):
    return {'index': index, 'count': 0, 'value': 0, 'mean_value': 0, 'p': p, 'subtree': {}, 'parent': parent,
            'virtual_loss': 0}


def new_subtree(policy, board, parent, add_noise=False):
    """play.py:391-421: one child per legal action, ascending, prior = raw network output (not renormalised);
    with add_noise the priors are mixed with Dirichlet noise drawn over ALL actions."""
    import numpy.ma as ma
    priors = ma.masked_array(policy, mask=legal_moves(board)).reshape(-1)
    if add_noise:
        noise = np.random.dirichlet([conf['DIRICHLET_ALPHA']] * priors.shape[0])
        priors = (1 - conf['DIRICHLET_EPSILON']) * priors + conf['DIRICHLET_EPSILON'] * noise
    return {move: _new_node(move, p, parent) for move, p in enumerate(priors) if p is not ma.masked}


def new_tree(policy, board, add_noise=False):
    root = _new_node(-1, 1, None)
    root['subtree'] = new_subtree(policy, board, parent=root, add_noise=add_noise)
    return root


def _puct(subtree):
    """(action, child, Q + U) for every child, in dict order; U = Cpuct * p * sqrt(sum N) / (1 + N)."""
    from math import sqrt
    total_n = sqrt(sum(child['count'] for child in subtree.values())) or 1
    for action, child in subtree.items():
        yield action, child, child['mean_value'] + Cpuct * child['p'] * total_n / (1. + child['count'])


def top_one_with_virtual_loss(node):
    """play.py:308-323: best non-busy child, first index wins ties, {} when every child is busy."""
    best, best_value = {}, -100
    for action, child, value in _puct(node['subtree']):
        if child.get('virtual_loss', 0) > 0:
            continue
        if value > best_value:
            best, best_value = {'action': action, 'node': child}, value
    return best


def top_one_action(subtree):
    """play.py:325-336."""
    best = {'action': -1, 'value': -1, 'node': None}
    for action, child, value in _puct(subtree):
        if value > best['value']:
            best = {'action': action, 'value': value, 'node': child}
    return best


def top_n_actions(subtree, top_n):
    """play.py:338-352: the top_n children by Q + U, descending, earlier index first among equals."""
    ranked = []
    for action, child, value in _puct(subtree):
        if len(ranked) < top_n or value > ranked[-1]['value']:
            pos = len(ranked)
            while pos > 0 and ranked[pos - 1]['value'] < value:
                pos -= 1
            ranked.insert(pos, {'action': action, 'value': value, 'node': child})
            del ranked[top_n:]
    return ranked


def tree_depth(tree):
    if tree['subtree'] is None:
        return 1
    return 1 + max([tree_depth(child) for child in tree['subtree'].values()] or [0])
