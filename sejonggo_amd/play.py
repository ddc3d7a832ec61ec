"""Drop-in mirror of the reference's play.py (rules part) running on MI355X through libsgo_hip.so.

Same names, argument meaning, in-place mutation and error behaviour as the reference:
  game_init (play.py:295-299), make_play (:226-242), legal_moves (:71-104), get_winner (:274-284),
  index2coord / coord2index / gtpcoord2index / str_coord (:22-43), get_surrounding (:45-55), get_liberties (:57-69),
  get_real_board (:106-112), show_board (:114-147), capture_group (:159-180), take_stones (:182-217),
  swap_player (:219-224), color_board / _get_points (:244-292), choose_first_player (:301-306).
Every flood fill (captures, liberties, territory colouring, legality, scoring) runs in libsgo_hip.so; what stays on
the host is format conversion and the ORDER in which capture_group lists the stones it was told are captured.
Boards are caller-owned numpy int32 arrays [1,S,S,17] (batches [n,S,S,17] are accepted too and are
processed in one kernel launch).  SIZE / KOMI are read from sejonggo_amd.conf at call time unless the
board's own shape says otherwise.
"""
import ctypes as C
from random import random

import numpy as np

from . import _lib
from .conf import conf

SWAP_INDEX = [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14]
colstr = 'ABCDEFGHJKLMNOPQRST'
dxdys = [(1, 0), (-1, 0), (0, 1), (0, -1)]


def _size_of(board):
    return board.shape[-2]


def str_coord(c, size=None):
    """play.py:22-28 (michi-style padded-board coordinate -> "D4"; "resign" and pass are spelled out)."""
    S = size or conf['SIZE']
    if c == "resign":
        return c
    if S * S == c:
        return 'pass'
    row, col = divmod(c - (S + 3), S + 2)
    return '%c%d' % (colstr[col], S - row)


def gtpcoord2index(x, y, size=None):
    S = size or conf['SIZE']
    return S * (y - 1) + (x - 1)


def get_surrounding(x, y, size=None):
    """play.py:45-55: the on-board neighbours in the order up, right, down, left."""
    S = size or conf['SIZE']
    out = []
    if y - 1 >= 0:
        out.append((x, y - 1))
    if x + 1 < S:
        out.append((x + 1, y))
    if y + 1 < S:
        out.append((x, y + 1))
    if x - 1 >= 0:
        out.append((x - 1, y))
    return out


def _embed(real_board):
    """A plain 2-D board of any shape h x w (the reference's helpers are called with 3x3 ... 9x9 arrays,
    test/tests.py:51-213) inside the smallest supported S x S board; the padding is wall (neither stone nor liberty)."""
    rb = np.asarray(real_board)
    h, w = rb.shape
    S = next((s for s in _lib.SUPPORTED_SIZES if s >= max(h, w)), None)
    if S is None:
        raise ValueError("board of shape %s is larger than 19x19" % (rb.shape,))
    cells = np.full((1, S, S), 2, dtype=np.int8)
    cells[0, :h, :w] = np.clip(rb, -2, 2)
    cells[0, :h, :w][(rb != 0) & (rb != 1) & (rb != -1)] = 2
    return cells, S, h, w


def _query(mode, cells, S, x, y, color):
    lib = _lib.require_gpu()
    member = np.zeros((1, S, S), dtype=np.uint8)
    liberty = np.zeros((1, S, S), dtype=np.uint8)
    xs, ys, cs = (np.array([v], dtype=np.int32) for v in (x, y, color))
    _lib.check(lib.sgo_board_query(C.c_int(S), C.c_int(1), C.c_int(mode), _lib.ptr(cells), _lib.ptr(xs), _lib.ptr(ys),
                                   _lib.ptr(cs), _lib.ptr(member), _lib.ptr(liberty)), "sgo_board_query")
    return member[0].astype(bool), liberty[0].astype(bool)


def capture_group(x, y, real_board, group=None):
    """play.py:159-180: None when the group of the stone at (x, y) touches an empty point, otherwise its stones --
    listed in the reference's depth-first order (directions right, left, down, up; test/tests.py:146-199 pins it).
    Whether the group lives is decided on the GPU (k_board_query); the host only orders the member set."""
    cells, S, h, w = _embed(real_board)
    if not (0 <= x < w and 0 <= y < h):
        raise IndexError("capture_group: (%d, %d) outside a %dx%d board" % (x, y, h, w))
    color = int(cells[0, y, x])
    if color not in (-1, 0, 1):
        color = 0
    member, liberty = _query(0, cells, S, x, y, color)
    if liberty.any():
        return None
    if group is None:
        group = [(x, y)]
    seen = set(group)
    stack = [(x, y, 0)]
    while stack:
        cx, cy, d = stack.pop()
        while d < 4:
            nx, ny = cx + dxdys[d][0], cy + dxdys[d][1]
            d += 1
            if (nx, ny) in seen or not (0 <= nx < w and 0 <= ny < h) or not member[ny, nx]:
                continue
            seen.add((nx, ny))
            group.append((nx, ny))
            stack.append((cx, cy, d))
            stack.append((nx, ny, 0))
            break
    return group


def get_liberties(x, y, board, color=None, parent_x=None, parent_y=None):
    """play.py:57-69: empty points next to (x, y) or next to the stones of `color` connected to it.  The reference walks
    the group recursively and only skips the stone it came from (so it never returns on a group that contains a cycle);
    here the set is one GPU flood fill.  (parent_x, parent_y), the recursion's own bookkeeping, excludes that point's
    side of the walk exactly as a first call with it would."""
    if color is None:
        color = board[0, 0, 0, -1]
    rb = np.array(get_real_board(board))
    if parent_x is not None and parent_y is not None and 0 <= parent_y < rb.shape[0] and 0 <= parent_x < rb.shape[1]:
        rb[parent_y, parent_x] = 2          # neither a liberty nor part of the group for this walk
    cells, S, h, w = _embed(rb)
    _, liberty = _query(0, cells, S, x, y, int(color))
    ys, xs = np.nonzero(liberty[:h, :w])
    return [(int(a), int(b)) for a, b in zip(xs, ys)]


def take_stones(x, y, board):
    """play.py:182-217, in place on planes 0/1 of a board tensor; returns the board."""
    lib = _lib.require_gpu()
    if board.dtype != np.int32 or not board.flags['C_CONTIGUOUS']:
        raise TypeError("board must be a C-contiguous int32 array (the reference's game_init dtype)")
    S = _size_of(board)
    xs, ys = np.array([x], dtype=np.int32), np.array([y], dtype=np.int32)
    _lib.check(lib.sgo_take_stones(C.c_int(S), C.c_int(1), _lib.ptr(board), _lib.ptr(xs), _lib.ptr(ys)), "sgo_take_stones")
    return board


def swap_player(board):
    """play.py:219-224: swap the plane pairs and negate the colour plane, in place; returns the new side to move.
    (A pure permutation of the caller's array: nothing to compute.)"""
    player = board[0, 0, 0, -1]
    board[:, :, :, range(16)] = board[:, :, :, SWAP_INDEX]
    player = -1 if player == 1 else 1
    board[:, :, :, -1] = player
    return player


def color_board(real_board, color):
    """play.py:262-271: a copy of `real_board` with every empty region that touches a stone of `color` painted in
    that colour (the fill itself: k_board_query mode 1)."""
    out = np.copy(real_board)
    cells, S, h, w = _embed(real_board)
    member, _ = _query(1, cells, S, 0, 0, int(color))
    out[member[:h, :w]] = color
    return out


def _get_points(real_board):
    """play.py:286-292: {value: count} of color_board(+1) + color_board(-1): 1 / -1 territory, 2 / -2 stones, 0 neutral."""
    total = color_board(real_board, 1) + color_board(real_board, -1)
    unique, counts = np.unique(total, return_counts=True)
    return dict(zip(unique, counts))


def choose_first_player(model1, model2):
    """play.py:301-306: one draw of Python's `random`; model1 moves first when it is below .5."""
    if random() < .5:
        return model1, model2
    return model2, model1


def _show_board(board, policy):
    real_board = get_real_board(board)
    size = real_board.shape[0]
    x = y = None
    if policy is not None:
        y, x = divmod(int(np.argmax(policy)), size)
    rows = []
    for j, row in enumerate(real_board):
        rows.append("".join(u"\u25cb " if c == 1 else u"\u25cf " if c == -1 else
                            u"X " if (policy is not None and i == x and j == y) else u". " for i, c in enumerate(row)) + "\n")
    text = "".join(rows)
    if policy is not None and y == size:
        text += "Pass policy"
    return text


def show_board(board, policy=None, history=1):
    """play.py:139-147: the position (and optionally the `history` previous ones, oldest first) as text."""
    out = []
    for i in reversed(range(history)):
        tmp = np.copy(board)[:, :, :, i:]
        if i % 2 == 1:
            tmp[:, :, :, -1] *= -1
        out.append(_show_board(tmp, policy))
    return "\n".join(out)


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
