"""ctypes binding of oracle/libsgo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Every function cites the reference through the C function it wraps (oracle/sgo_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsgo_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "sgo_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.ora_game_new.restype = C.c_void_p
        _lib.ora_game_new.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        _lib.ora_game_tree_serialize.restype = C.c_size_t
        _lib.ora_game_tree_serialize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        for n in ("ora_game_free", "ora_game_set_draws", "ora_game_phase", "ora_game_error", "ora_game_pending",
                  "ora_game_submit", "ora_game_n_moves", "ora_game_move", "ora_game_result", "ora_game_counters",
                  "ora_game_board", "ora_game_move_n", "ora_game_root_table", "ora_game_set_resign", "ora_game_set_halt"):
            getattr(_lib, n).argtypes = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _b32(board):
    b = np.ascontiguousarray(board, dtype=np.int32)
    return b


def board_size(board):
    return board.shape[-2]


# ---------------------------------------------------------------- rules
def game_init(S):
    b = np.zeros((1, S, S, 17), dtype=np.int32)
    lib().ora_game_init(C.c_int(S), _p(b))
    return b, 1


def make_play(x, y, board, color=None):
    """In place on an int32 C-contiguous board [1,S,S,17]; returns (board, mover)."""
    assert board.dtype == np.int32 and board.flags["C_CONTIGUOUS"]
    S = board_size(board)
    r = lib().ora_make_play(C.c_int(S), _p(board), C.c_int(int(x)), C.c_int(int(y)), C.c_int(0 if color is None else int(color)))
    if r == -101:
        raise AssertionError("occupied")
    if r == -102:
        raise IndexError("out of range")
    return board, r


def legal_moves(board):
    b = _b32(board)
    S = board_size(b)
    m = np.zeros(S * S + 1, dtype=np.uint8)
    lib().ora_legal_moves(C.c_int(S), _p(b), _p(m))
    return m


def get_real_board(board):
    b = _b32(board)
    S = board_size(b)
    rb = np.zeros((S, S), dtype=np.int8)
    lib().ora_get_real_board(C.c_int(S), _p(b), _p(rb))
    return rb


def capture_group(x, y, real_board):
    rb = np.ascontiguousarray(real_board, dtype=np.int8)
    S = rb.shape[0]
    grp = np.zeros((S * S, 2), dtype=np.int16)
    n = lib().ora_capture_group(C.c_int(S), _p(rb), C.c_int(x), C.c_int(y), _p(grp))
    if n == 0:
        return None
    return [tuple(int(v) for v in g) for g in grp[:n]]


def color_board(real_board, color):
    rb = np.ascontiguousarray(real_board, dtype=np.int8)
    S = rb.shape[0]
    out = np.zeros_like(rb)
    lib().ora_color_board(C.c_int(S), _p(rb), C.c_int(color), _p(out))
    return out


def get_points(real_board):
    rb = np.ascontiguousarray(real_board, dtype=np.int8)
    S = rb.shape[0]
    cnt = np.zeros(5, dtype=np.int32)
    lib().ora_get_points(C.c_int(S), _p(rb), _p(cnt))
    return {v - 2: int(cnt[v]) for v in range(5) if cnt[v]}


def get_winner(board, komi=5.5):
    b = _b32(board)
    S = board_size(b)
    black = C.c_int(0)
    white = C.c_double(0)
    w = lib().ora_get_winner(C.c_int(S), _p(b), C.c_double(komi), C.byref(black), C.byref(white))
    return w, black.value, white.value


# ---------------------------------------------------------------- symmetry
def sym_lut(S, k):
    lut = np.zeros(S * S + 1, dtype=np.int32)
    lib().ora_sym_lut(C.c_int(S), C.c_int(k), _p(lut))
    return lut


def sym_board(k, boards):
    b = np.ascontiguousarray(boards, dtype=np.int32)
    S = b.shape[-2]
    out = np.zeros_like(b)
    flat_in = b.reshape(-1, S, S, 17)
    flat_out = out.reshape(-1, S, S, 17)
    for i in range(flat_in.shape[0]):
        lib().ora_sym_board(C.c_int(S), C.c_int(k), _p(flat_in[i]), _p(flat_out[i]))
    return out


def sym_policy_inverse(S, k, policy):
    p = np.ascontiguousarray(policy, dtype=np.float32).reshape(-1, S * S + 1)
    out = np.zeros_like(p)
    for i in range(p.shape[0]):
        lib().ora_sym_policy_inverse(C.c_int(S), C.c_int(k), _p(p[i]), _p(out[i]))
    return out.reshape(np.shape(policy))


# ---------------------------------------------------------------- selectors on flat child tables
def top_one_with_virtual_loss(P, N, Q, V, EX, f64):
    A = len(P)
    return lib().ora_top_one_with_virtual_loss(
        C.c_int(A), _p(np.ascontiguousarray(P, np.float64)), _p(np.ascontiguousarray(N, np.int32)),
        _p(np.ascontiguousarray(Q, np.float32)), _p(np.ascontiguousarray(V, np.int8)),
        _p(np.ascontiguousarray(EX, np.int8)), C.c_int(int(f64)))


def top_one_action(P, N, Q, EX, f64):
    A = len(P)
    return lib().ora_top_one_action(
        C.c_int(A), _p(np.ascontiguousarray(P, np.float64)), _p(np.ascontiguousarray(N, np.int32)),
        _p(np.ascontiguousarray(Q, np.float32)), _p(np.ascontiguousarray(EX, np.int8)), C.c_int(int(f64)))


def top_n_actions(P, N, Q, EX, f64, top_n):
    A = len(P)
    out = np.full(top_n, -1, dtype=np.int32)
    n = lib().ora_top_n_actions(
        C.c_int(A), _p(np.ascontiguousarray(P, np.float64)), _p(np.ascontiguousarray(N, np.int32)),
        _p(np.ascontiguousarray(Q, np.float32)), _p(np.ascontiguousarray(EX, np.int8)), C.c_int(int(f64)),
        C.c_int(top_n), _p(out))
    return out[:n]


# ---------------------------------------------------------------- async self-play game
PH_ROOT, PH_LEAF, PH_DONE = 0, 1, 2


class Game(object):
    """State machine restating play_game_async (nomodel_self_play.py:142-271) for one game.

    Drive it with ``pending()`` -> evaluate boards -> ``submit(policies, values)`` until
    ``phase == PH_DONE``; or call ``run(net)``.
    """

    def __init__(self, S, sims, energy, stop_exploration, num_moves=None, self_play=True, komi=5.5,
                 dir_eps=0.25, uniforms=None, noises=None, resign=None, halt_at=None):
        self.S, self.A = S, S * S + 1
        self._g = C.c_void_p(lib().ora_game_new(C.c_int(S), C.c_double(komi), C.c_int(sims), C.c_int(energy),
                                                C.c_int(stop_exploration), C.c_int(-1 if num_moves is None else num_moves),
                                                C.c_int(1 if self_play else 0), C.c_double(dir_eps)))
        if not self._g:
            raise ValueError("bad game parameters")
        self._u = np.ascontiguousarray(uniforms if uniforms is not None else np.zeros(0), dtype=np.float64)
        self._n = np.ascontiguousarray(noises if noises is not None else np.zeros((0, self.A)), dtype=np.float64).reshape(-1, self.A)
        lib().ora_game_set_draws(self._g, _p(self._u), C.c_int(len(self._u)), _p(self._n), C.c_int(len(self._n)))
        if resign is not None:
            lib().ora_game_set_resign(self._g, C.c_int(1), C.c_float(resign))
        if halt_at is not None:
            lib().ora_game_set_halt(self._g, C.c_int(halt_at))
        self._buf = np.zeros((64, S, S, 17), dtype=np.int32)

    def __del__(self):
        try:
            lib().ora_game_free(self._g)
        except Exception:
            pass

    @property
    def phase(self):
        return lib().ora_game_phase(self._g)

    @property
    def error(self):
        return lib().ora_game_error(self._g)

    def pending(self):
        n = lib().ora_game_pending(self._g, _p(self._buf))
        return self._buf[:n]

    def submit(self, policies, values):
        p = np.ascontiguousarray(policies, dtype=np.float32)
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        lib().ora_game_submit(self._g, _p(p), _p(v))

    def run(self, net, on_move=None):
        last = 0
        while self.phase != PH_DONE:
            boards = self.pending()
            p, v = net.predict_on_batch(boards)
            self.submit(p, v)
            if on_move is not None and (self.n_moves != last or self.phase == PH_DONE):
                last = self.n_moves
                on_move(self)
        if self.error:
            raise RuntimeError("oracle game error %d" % self.error)
        return self

    @property
    def n_moves(self):
        return lib().ora_game_n_moves(self._g)

    def move(self, i):
        a, pl, v = C.c_int(0), C.c_int(0), C.c_float(0)
        board = np.zeros((1, self.S, self.S, 17), dtype=np.int32)
        pol = np.zeros(self.A, dtype=np.float64)
        lib().ora_game_move(self._g, C.c_int(i), C.byref(a), C.byref(pl), C.byref(v), _p(board), _p(pol))
        return {"action": a.value, "player": pl.value, "value": np.float32(v.value), "board": board, "policy": pol}

    def result(self):
        w, b, wh, er, lp = C.c_int(0), C.c_int(0), C.c_double(0), C.c_int(0), C.c_int(0)
        lib().ora_game_result(self._g, C.byref(w), C.byref(b), C.byref(wh), C.byref(er), C.byref(lp))
        return {"winner": w.value, "black": b.value, "white": wh.value, "end_reason": er.value, "last_player": lp.value}

    def counters(self):
        a, b, c = C.c_long(0), C.c_long(0), C.c_long(0)
        lib().ora_game_counters(self._g, C.byref(a), C.byref(b), C.byref(c))
        return {"n_predict": a.value, "n_root_predict": b.value, "none_events": c.value}

    def board(self):
        b = np.zeros((1, self.S, self.S, 17), dtype=np.int32)
        lib().ora_game_board(self._g, _p(b))
        return b

    def root_table(self):
        A = self.A
        N = np.zeros(A, np.int32); W = np.zeros(A, np.float32); Q = np.zeros(A, np.float32)
        P = np.zeros(A, np.float64); EX = np.zeros(A, np.int8)
        rc, rv = C.c_int32(0), C.c_float(0)
        lib().ora_game_root_table(self._g, _p(N), _p(W), _p(Q), _p(P), _p(EX), C.byref(rc), C.byref(rv))
        return {"N": N, "W": W, "Q": Q, "P": P, "EX": EX, "root_count": rc.value, "root_value": np.float32(rv.value)}

    def tree_serialize(self):
        nn, ne = C.c_long(0), C.c_long(0)
        sz = lib().ora_game_tree_serialize(self._g, None, C.c_size_t(0), C.byref(nn), C.byref(ne))
        buf = np.zeros(sz, dtype=np.uint8)
        lib().ora_game_tree_serialize(self._g, _p(buf), C.c_size_t(sz), C.byref(nn), C.byref(ne))
        return buf, nn.value, ne.value
