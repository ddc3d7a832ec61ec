#!/usr/bin/env python3
"""Golden-vector generator: runs the *Python reference* (drsagitn/sejonggo, mounted read-only at
/root/reference) in THIS container and records inputs + expected outputs as small .npz fixtures.

The reference cannot travel to the GPU box, so everything the parity tests need is captured here
as plain arrays (np.load(allow_pickle=False) reads them).  Run from anywhere:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen_golden.py            # all fixtures
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen_golden.py --only sym # one family

Recipe: cwd is a scratch dir (so the reference's logconfig.json is not picked up),
`sgfsave` and `model` are stubbed in sys.modules (they need sgfmill/h5py/TensorFlow, absent here),
conf[...] is set BEFORE play/self_play/nomodel_self_play are imported (constants are captured at
import), one subprocess per board size.  RNG streams of the reference (np.random.choice,
np.random.dirichlet, random.choice) cannot be matched by a GPU implementation, so the harness
replaces them with recorded draws: the draws are part of the fixture ("RNG stream parity unpinned;
draws injected").  The multiprocessing Pool / SimpleQueue of simulation_workers.py are replaced by
in-process fakes whose results arrive in launch order (one legal interleaving of the reference's
racy pool); pickling is emulated with deepcopy / np.copy.
"""
import argparse
import hashlib
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# ----------------------------------------------------------------------------------------------
# child-process side: everything below runs with the reference importable
# ----------------------------------------------------------------------------------------------
def _setup_reference(size, sims, energy, komi=5.5):
    import types
    sys.setrecursionlimit(20000)
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    sgfsave = types.ModuleType("sgfsave")
    for n in ("save_self_play_data", "save_game_data", "save_game_sgf", "save_file"):
        setattr(sgfsave, n, lambda *a, **k: None)
    sys.modules["sgfsave"] = sgfsave
    model = types.ModuleType("model")
    model.load_best_model = lambda *a, **k: None
    model.load_latest_model = lambda *a, **k: None
    sys.modules["model"] = model
    from conf import conf
    conf["SIZE"] = size
    conf["KOMI"] = komi
    conf["MCTS_SIMULATIONS"] = sims
    conf["ENERGY"] = energy
    conf["N_GAME_PROCESS"] = 1
    conf["SHOW_EACH_MOVE"] = False
    conf["SHOW_END_GAME"] = False
    return conf


def _sha8(arr):
    import numpy as np
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(arr).tobytes()).digest()[:8], dtype=np.uint8)


def _board_i8(board):
    import numpy as np
    return board[0].astype(np.int8)


SCRIPTED_9 = {
    # name: list of (x, y, color) ; color 0 = None (to-play).  y == 9 is a pass.
    # Situations mirror the ones the reference pins in test/tests.py:215-481 (self-suicide,
    # capture-is-not-suicide, suicide, ko, two-stone capture is not ko, corner captures).
    "self_suicide": [(0, 0, 0), (1, 0, 0), (8, 9, 0), (2, 1, 0), (8, 8, 0), (3, 0, 0), (2, 0, 0)],
    "capture_not_suicide": [(0, 0, 0), (1, 0, 0), (1, 1, 0), (2, 1, 0), (8, 8, 0), (3, 0, 0), (2, 0, 0)],
    "suicide_illegal": [(0, 1, 0), (1, 0, 0), (1, 1, 0), (2, 1, 0), (8, 8, 0), (3, 0, 0)],
    "ko": [(1, 0, 0), (2, 0, 0), (0, 1, 0), (3, 1, 0), (1, 2, 0), (2, 2, 0), (2, 1, 0), (1, 1, 0),
           (8, 8, 0), (7, 7, 0), (2, 1, 0)],
    "two_stone_capture": [(2, 0, 0), (0, 0, 0), (2, 1, 0), (1, 0, 0), (0, 2, 0), (0, 1, 0), (1, 2, 0),
                          (1, 1, 0), (8, 8, 0), (8, 9, 0), (7, 7, 0), (8, 9, 0), (2, 2, 0), (8, 9, 0)],
    "explicit_colors": [(2, 1, 1), (2, 0, 1), (3, 1, -1), (1, 1, 1), (4, 1, -1), (2, 2, -1), (3, 0, -1),
                        (3, 2, 1), (8, 9, -1), (8, 9, 1), (0, 0, 1), (0, 1, -1), (1, 0, -1)],
    "corner_eye": [(1, 0, 0), (8, 8, 0), (0, 1, 0), (8, 7, 0), (1, 1, 0), (0, 0, 0), (7, 8, 0), (0, 0, 0)],
    "double_pass": [(4, 4, 0), (8, 9, 0), (8, 9, 0), (3, 3, 0)],
}


def child_rules(size, out):
    """Random + scripted playouts through make_play / legal_moves / get_winner."""
    import numpy as np
    _setup_reference(size, 8, 8)
    import play
    S, A = size, size * size + 1
    seed_off = int(os.environ.get("SGO_GOLDEN_SEED", "0"))
    rng = np.random.RandomState(1234 + size + seed_off)
    data = {}
    games = []
    if size == 9 and seed_off == 0:
        for name in sorted(SCRIPTED_9):
            games.append(("scripted_" + name, SCRIPTED_9[name]))
    n_random = {5: 6, 7: 4, 9: 8, 13: 3, 19: 5}[size] if seed_off == 0 else {5: 4, 7: 3, 9: 4, 13: 2, 19: 1}[size]
    for g in range(n_random):
        games.append(("random_%d" % g, None))
    for gi, (name, script) in enumerate(games):
        board, _ = play.game_init()
        moves, players, masks, hashes, fulls, full_at, winners = [], [], [], [], [], [], []
        masks.append(np.packbits(play.legal_moves(board).astype(np.uint8)))
        hashes.append(_sha8(board))
        max_plies = len(script) if script is not None else int(rng.randint(S * S // 2, 2 * S * S))
        use_colors = script is None and (gi % 4 == 3)
        for ply in range(max_plies):
            if script is not None:
                x, y, color = script[ply]
            else:
                mask = play.legal_moves(board)
                r = rng.rand()
                empties = [i for i in range(S * S)
                           if board[0, i // S, i % S, 0] == 0 and board[0, i // S, i % S, 1] == 0]
                legal = [i for i in range(S * S) if mask[i] == 0]
                if r < 0.04 or not empties:
                    a = S * S
                elif r < 0.16 or not legal:
                    a = empties[rng.randint(len(empties))]  # may be a suicide / ko-violating move
                else:
                    a = legal[rng.randint(len(legal))]
                x, y = play.index2coord(a)
                color = 0
                if use_colors and rng.rand() < 0.3:
                    color = int(rng.choice([-1, 1]))
            b2, mover = play.make_play(x, y, board, None if color == 0 else color)
            assert b2 is board
            moves.append((x, y, color))
            players.append(int(mover))
            masks.append(np.packbits(play.legal_moves(board).astype(np.uint8)))
            hashes.append(_sha8(board))
            if ply % 16 == 15 or ply == max_plies - 1:
                fulls.append(_board_i8(board))
                full_at.append(ply + 1)
                w, bp, wp = play.get_winner(board)
                winners.append((int(w), int(bp), float(wp)))
        p = "g%02d_" % gi
        data[p + "moves"] = np.array(moves, dtype=np.int16).reshape(-1, 3)
        data[p + "players"] = np.array(players, dtype=np.int8)
        data[p + "masks"] = np.array(masks, dtype=np.uint8)
        data[p + "hashes"] = np.array(hashes, dtype=np.uint8)
        data[p + "fulls"] = np.array(fulls, dtype=np.int8).reshape(-1, S, S, 17)
        data[p + "full_at"] = np.array(full_at, dtype=np.int32)
        data[p + "winners"] = np.array(winners, dtype=np.float64).reshape(-1, 3)
        data[p + "name"] = np.frombuffer(name.encode(), dtype=np.uint8)
    data["n_games"] = np.array(len(games), dtype=np.int32)
    data["size"] = np.array(size, dtype=np.int32)
    data["komi"] = np.array(5.5)
    np.savez_compressed(out, **data)


def child_sgf(out):
    """Replays of the reference's own 19x19 fixtures real_games/*.sgf (moves become data)."""
    import numpy as np
    size = 19
    _setup_reference(size, 8, 8)
    import play
    data = {}
    names = sorted(os.listdir(os.path.join(REF, "real_games")))
    for gi, fn in enumerate(names):
        txt = open(os.path.join(REF, "real_games", fn)).read()
        seq = re.findall(r";([BW])\[([a-s]{0,2})\]", txt)
        board, _ = play.game_init()
        moves, masks, hashes, fulls, full_at, winners = [], [], [], [], [], []
        masks.append(np.packbits(play.legal_moves(board).astype(np.uint8)))
        hashes.append(_sha8(board))
        for ply, (c, co) in enumerate(seq):
            if co == "":
                x, y = 0, size
            else:
                x, y = ord(co[0]) - 97, ord(co[1]) - 97
            color = 1 if c == "B" else -1
            play.make_play(x, y, board, color)
            moves.append((x, y, color))
            masks.append(np.packbits(play.legal_moves(board).astype(np.uint8)))
            hashes.append(_sha8(board))
            if ply % 16 == 15 or ply == len(seq) - 1:
                fulls.append(_board_i8(board))
                full_at.append(ply + 1)
                w, bp, wp = play.get_winner(board)
                winners.append((int(w), int(bp), float(wp)))
        p = "g%02d_" % gi
        data[p + "moves"] = np.array(moves, dtype=np.int16)
        data[p + "masks"] = np.array(masks, dtype=np.uint8)
        data[p + "hashes"] = np.array(hashes, dtype=np.uint8)
        data[p + "fulls"] = np.array(fulls, dtype=np.int8)
        data[p + "full_at"] = np.array(full_at, dtype=np.int32)
        data[p + "winners"] = np.array(winners, dtype=np.float64)
        data[p + "name"] = np.frombuffer(fn.encode(), dtype=np.uint8)
        data[p + "final_sha1"] = np.frombuffer(hashlib.sha1(board.tobytes()).hexdigest()[:12].encode(), dtype=np.uint8)
    data["n_games"] = np.array(len(names), dtype=np.int32)
    data["size"] = np.array(size, dtype=np.int32)
    data["komi"] = np.array(5.5)
    np.savez_compressed(out, **data)


def child_sym(size, out):
    import numpy as np
    _setup_reference(size, 8, 8)
    import symmetry as sy
    S, A = size, size * size + 1
    # canonical order used by this repo: 0 id, 1 left_diagonal, 2 vertical_axis, 3 horizontal_axis,
    # 4 rot90, 5 rot180, 6 rot270 (= symmetry.SYMMETRIES order, symmetry.py:117-125), 7 right_diagonal.
    fwd = [sy._id, sy.left_diagonal, sy.vertical_axis, sy.horizontal_axis, sy.rotation_90,
           sy.rotation_180, sy.rotation_270, sy.right_diagonal]
    rev = [sy._id, sy.reverse_left_diagonal, sy.reverse_vertical_axis, sy.reverse_horizontal_axis,
           sy.reverse_rotation_90, sy.reverse_rotation_180, sy.reverse_rotation_270,
           sy.reverse_right_diagonal]
    luts = [list(range(A)), sy.LEFT_DIAGONAL_SWAP, sy.VERTICAL_AXIS_SWAP, sy.HORIZONTAL_AXIS_SWAP,
            sy.ROTATION_90_SWAP, sy.ROTATION_180_SWAP, sy.ROTATION_270_SWAP, sy.RIGHT_DIAGONAL_SWAP]
    assert [f for f, _ in sy.SYMMETRIES] == fwd[:7]
    rng = np.random.RandomState(77 + size)
    boards = rng.randint(-1, 2, size=(3, S, S, 17)).astype(np.int32)
    policy = rng.rand(3, A).astype(np.float32)
    data = {"size": np.array(size, dtype=np.int32), "boards": boards.astype(np.int8), "policy": policy,
            "luts": np.array(luts, dtype=np.int32)}
    for k in range(8):
        tb = np.array(fwd[k](np.copy(boards)))
        data["fwd%d" % k] = tb.astype(np.int8)
        data["rev%d" % k] = np.array(rev[k](np.copy(policy)))
        # round trip the reference relies on: a net that is equivariant sees policy transformed like the board
    np.savez_compressed(out, **data)


def child_puct(out):
    """Unit-level selectors: top_one_with_virtual_loss / top_one_action / top_n_actions on random
    child tables, in the two dtype regimes the reference produces under numpy 2 (float32 priors from
    the net; float64 priors after Dirichlet mixing at a fresh root)."""
    import numpy as np
    _setup_reference(9, 8, 8)
    import play
    rng = np.random.RandomState(4242)
    A = 82
    n_cases = 600
    P = np.zeros((n_cases, A), dtype=np.float64)
    N = np.zeros((n_cases, A), dtype=np.int32)
    Q = np.zeros((n_cases, A), dtype=np.float32)
    V = np.zeros((n_cases, A), dtype=np.int8)
    EX = np.zeros((n_cases, A), dtype=np.int8)
    F64 = np.zeros((n_cases,), dtype=np.int8)
    out_vl = np.zeros((n_cases,), dtype=np.int32)
    out_one = np.zeros((n_cases,), dtype=np.int32)
    out_top = np.full((n_cases, 8), -1, dtype=np.int32)
    for c in range(n_cases):
        f64 = c % 3 == 2
        F64[c] = f64
        exist = rng.rand(A) < rng.choice([0.05, 0.3, 0.9, 1.0])
        if not exist.any():
            exist[rng.randint(A)] = True
        levels = rng.choice([3, 10, 1000])
        hi = int(rng.choice([1, 4, 60, 2000]))
        subtree = {}
        for a in range(A):
            if not exist[a]:
                continue
            if f64:
                p = np.float64(rng.randint(1, levels + 1)) / np.float64(levels * 7)
            else:
                p = np.float32(rng.randint(1, levels + 1)) / np.float32(levels * 7)
            n = int(rng.randint(0, hi)) if rng.rand() < 0.6 else 0
            if n:
                value = np.float32(0)
                for _ in range(min(n, 5)):
                    value = value + np.float32(rng.uniform(-1, 1))
                q = value / float(n)
                assert type(q) is np.float32
            else:
                q = 0
            vl = 2 if rng.rand() < rng.choice([0.0, 0.2, 0.97]) else 0
            subtree[a] = {"index": a, "count": n, "value": 0, "mean_value": q, "p": p, "subtree": {},
                          "parent": None, "virtual_loss": vl}
            P[c, a], N[c, a], Q[c, a], V[c, a], EX[c, a] = p, n, q, vl, 1
        r = play.top_one_with_virtual_loss({"subtree": subtree})
        out_vl[c] = r["action"] if r else -1
        r = play.top_one_action(subtree)
        out_one[c] = r["action"]
        r = play.top_n_actions(subtree, 8)
        for i, d in enumerate(r):
            out_top[c, i] = d["action"]
    np.savez_compressed(out, P=P, N=N, Q=Q, V=V, EX=EX, F64=F64, out_vl=out_vl, out_one=out_one, out_top=out_top)


def _tree_hash(root):
    """Canonical serialisation: pre-order, ascending action; per child
    <i action, i count, f value, f mean_value, d p, i virtual_loss, i expanded>."""
    import numpy as np
    h = hashlib.sha1()
    n_nodes = 0
    n_expanded = 0
    stack = [root]
    # explicit pre-order with ascending children
    def rec(node):
        nonlocal n_nodes, n_expanded
        for a in node["subtree"]:
            c = node["subtree"][a]
            for k in ("value", "mean_value"):
                assert isinstance(c[k], (int, np.float32)), (k, type(c[k]))
            assert isinstance(c["p"], (np.float32, np.float64)), type(c["p"])
            h.update(struct.pack("<iiffdii", int(a), int(c["count"]), float(c["value"]), float(c["mean_value"]),
                                 float(c["p"]), int(c["virtual_loss"]), 1 if c["subtree"] else 0))
            n_nodes += 1
            if c["subtree"]:
                n_expanded += 1
                rec(c)
    prev = list(root["subtree"].keys())
    assert prev == sorted(prev)
    rec(root)
    return h.digest()[:16], n_nodes, n_expanded


def child_async(size, sims, energy, net_kind, num_moves, stop_exploration, seed, out):
    """Full play_game_async games (nomodel_self_play.py:142) with a deterministic stub net."""
    import copy
    import collections
    import numpy as np
    conf = _setup_reference(size, sims, energy)
    conf["STOP_EXPLORATION"] = stop_exploration
    import play
    import symmetry
    import predicting_queue_worker as pq
    import simulation_workers as sw
    import nomodel_self_play as ns
    from sejonggo_amd.stub_nets import make_stub
    S, A = size, size * size + 1
    # "a+b": a two-model evaluation game (evaluate_worker.py:137): BEST* requests go to net a, LATEST* to net b
    two = "+" in net_kind
    kinds = net_kind.split("+") if two else [net_kind, net_kind]
    nets = {"BEST": make_stub(kinds[0], size), "LATEST": make_stub(kinds[1], size)}
    net = nets["BEST"]
    counters = {"predict": 0, "root": 0, "by_model": {"BEST": 0, "LATEST": 0}}

    def which(indicator):
        return "BEST" if indicator.startswith("BEST") else "LATEST"

    def stub_predict(indicator, board, response_now=False):
        counters["predict"] += 1
        counters["by_model"][which(indicator)] += 1
        if response_now:
            counters["root"] += 1
        p, v = nets[which(indicator)].predict_on_batch(np.asarray(board))
        assert p.dtype == np.float32 and v.dtype == np.float32
        return p[0], v[0][0]

    for m in (pq, sw, ns):
        m.put_predict_request = stub_predict
        m.put_name_request = lambda ind: nets[which(ind)].name
    first_draw = 0.25 if seed % 2 == 0 else 0.75       # play.choose_first_player: random() < .5 => model1 moves first
    play.random = lambda: first_draw

    class FakePool(object):
        def apply_async(self, fn, args, error_callback=None, callback=None):
            leaf, board, moves, ind, orig, pid = args
            fn(copy.deepcopy(leaf), np.copy(board), list(moves), ind, orig, pid)  # emulate pickling

        def close(self):
            pass

        def join(self):
            pass

    class FakeQueue(object):
        def __init__(self):
            self.q = collections.deque()

        def put(self, x):
            self.q.append(x)

        def get(self):
            if not self.q:
                raise RuntimeError("reference would block forever: result queue empty")
            return self.q.popleft()

    sw.process_pool = FakePool()
    sw.simulation_result_queue[0] = FakeQueue()

    draw_rng = np.random.RandomState(seed)
    rec = {"uniforms": [], "noises": [], "none_events": 0}

    def fake_choice(moves, size=1, p=None):
        u = draw_rng.random_sample()
        rec["uniforms"].append(u)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        idx = int(np.searchsorted(cdf, u, side="right"))
        return [moves[idx]]

    def fake_dirichlet(alpha):
        g = draw_rng.gamma(alpha[0], size=len(alpha))  # any draw is fine: it is recorded
        noise = g / g.sum()
        rec["noises"].append(noise.astype(np.float64))
        return noise

    np.random.choice = fake_choice
    np.random.dirichlet = fake_dirichlet

    per_move = collections.defaultdict(list)
    orig_select = ns.select_play

    def wrapped_select(board, energy_, tree, temperature, indicator, gpuid):
        a = orig_select(board, energy_, tree, temperature, indicator, gpuid)
        n = np.zeros(A, dtype=np.int32)
        w = np.zeros(A, dtype=np.float32)
        q = np.zeros(A, dtype=np.float32)
        p = np.zeros(A, dtype=np.float64)
        ex = np.zeros(A, dtype=np.int8)
        for mv, d in tree["subtree"].items():
            n[mv], w[mv], q[mv], p[mv], ex[mv] = d["count"], d["value"], d["mean_value"], d["p"], 1
        hsh, n_nodes, n_exp = _tree_hash(tree)
        per_move["N"].append(n); per_move["W"].append(w); per_move["Q"].append(q)
        per_move["P"].append(p); per_move["EX"].append(ex)
        per_move["tree_hash"].append(np.frombuffer(hsh, dtype=np.uint8))
        per_move["n_nodes"].append(n_nodes); per_move["n_expanded"].append(n_exp)
        per_move["root_count"].append(int(tree["count"]))
        per_move["root_value"].append(np.float32(tree["value"]))
        per_move["action"].append(int(a))
        per_move["temperature"].append(int(temperature))
        per_move["model"].append(0 if which(indicator) == "BEST" else 1)
        return a

    ns.select_play = wrapped_select
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        if two:
            gd = ns.play_game_async("BEST_SYM", "LATEST_SYM", energy, stop_exploration, process_id=0, num_moves=num_moves)
        else:
            gd = ns.play_game_async("BEST_SYM", "BEST_SYM", energy, stop_exploration, process_id=0,
                                    self_play=True, num_moves=num_moves)
    rec["none_events"] = buf.getvalue().count("No best leaf")
    moves = gd["moves"]
    data = {
        "size": np.array(size, dtype=np.int32), "sims": np.array(sims, dtype=np.int32),
        "energy": np.array(energy, dtype=np.int32), "stop_exploration": np.array(stop_exploration, dtype=np.int32),
        "num_moves": np.array(-1 if num_moves is None else num_moves, dtype=np.int32),
        "net": np.frombuffer(net_kind.encode(), dtype=np.uint8),
        "komi": np.array(5.5),
        "uniforms": np.array(rec["uniforms"], dtype=np.float64),
        "noises": np.array(rec["noises"], dtype=np.float64).reshape(-1, A),
        "none_events": np.array(rec["none_events"], dtype=np.int32),
        "n_predict": np.array(counters["predict"], dtype=np.int32),
        "n_root_predict": np.array(counters["root"], dtype=np.int32),
        "move_index": np.array([mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S for mv in moves], dtype=np.int32),
        "move_xy": np.array([mv["move"] for mv in moves], dtype=np.int32).reshape(-1, 2),
        "move_player": np.array([mv["player"] for mv in moves], dtype=np.int8),
        "move_value": np.array([mv["value"] for mv in moves], dtype=np.float32),
        "move_policy": np.array([mv["policy"] for mv in moves], dtype=np.float64).reshape(-1, A),
        "move_board_hash": np.array([_sha8(mv["board"]) for mv in moves], dtype=np.uint8).reshape(-1, 8),
        "first_board": _board_i8(moves[0]["board"]) if moves else np.zeros((S, S, 17), np.int8),
        "last_board": _board_i8(moves[-1]["board"]) if moves else np.zeros((S, S, 17), np.int8),
        "winner": np.array(-99 if gd["winner"] is None else gd["winner"], dtype=np.int32),
        "result": np.frombuffer(gd["result"].encode(), dtype=np.uint8),
        "two_model": np.array(1 if two else 0, dtype=np.int32),
        "first_draw": np.array(first_draw),
        "modelB_name": np.frombuffer(gd["modelB_name"].encode(), dtype=np.uint8),
        "modelW_name": np.frombuffer(gd["modelW_name"].encode(), dtype=np.uint8),
        "winner_model": np.frombuffer(("" if gd["winner_model"] is None else gd["winner_model"]).encode(), dtype=np.uint8),
        "n_predict_best": np.array(counters["by_model"]["BEST"], dtype=np.int32),
        "n_predict_latest": np.array(counters["by_model"]["LATEST"], dtype=np.int32),
    }
    for k, v in per_move.items():
        data["pm_" + k] = np.array(v)
    np.savez_compressed(out, **data)
    print("async S=%d sims=%d E=%d net=%s: %d moves, result %s, %d predicts, none_events=%d" % (
        size, sims, energy, net_kind, len(moves), gd["result"], counters["predict"], rec["none_events"]))



def child_sync(size, sims, batch, net_kind, num_moves, stop_exploration, seed, out):
    """The sync path (self_play.py:28-163 simulate / mcts_decision / select_play, :164-290 play_game): the path the
    reference's own MCTS unit tests target.  THREAD_SIMULATION is switched off (same semantics, in-process branch)."""
    import collections
    import numpy as np
    conf = _setup_reference(size, sims, 8)
    conf["MCTS_BATCH_SIZE"] = batch
    conf["THREAD_SIMULATION"] = False
    conf["STOP_EXPLORATION"] = stop_exploration
    import play
    import symmetry
    symmetry.SYMMETRIES = symmetry.SYMMETRIES[0:1]      # identity only, as the reference's MCTSTestCase does
    import self_play as sp
    from sejonggo_amd.stub_nets import make_stub
    S, A = size, size * size + 1
    net = make_stub(net_kind, size)
    draw_rng = np.random.RandomState(seed)
    rec = {"uniforms": [], "noises": []}

    def fake_choice(moves, size=1, p=None):
        u = draw_rng.random_sample()
        rec["uniforms"].append(u)
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return [moves[int(np.searchsorted(cdf, u, side="right"))]]

    def fake_dirichlet(alpha):
        g = draw_rng.gamma(alpha[0], size=len(alpha))
        noise = g / g.sum()
        rec["noises"].append(noise.astype(np.float64))
        return noise

    np.random.choice = fake_choice
    np.random.dirichlet = fake_dirichlet
    data = {}
    # ---- (1) simulate() called repeatedly on one tree
    board, _ = play.game_init()
    for (x, y) in [(2, 2), (3, 2), (2, 3)][: 3 if size >= 5 else 0]:
        play.make_play(x, y, board)
    pol, val = net.predict_on_batch(board)
    tree = play.new_tree(pol[0], board, add_noise=False)
    hashes, counts = [], []
    for it in range(6):
        sp.simulate(tree, np.copy(board), net, batch, board[0, 0, 0, -1])
        h, nn, ne = _tree_hash(tree)
        hashes.append(np.frombuffer(h, dtype=np.uint8))
        counts.append((nn, ne, int(tree["count"])))
    data["sim_board"] = _board_i8(board)
    data["sim_hashes"] = np.array(hashes)
    data["sim_counts"] = np.array(counts, dtype=np.int64)
    data["sim_root_value"] = np.array(np.float32(tree["value"]))
    # ---- (2) a sync play_game
    per_move = collections.defaultdict(list)
    orig = sp.select_play

    def wrapped(policy, board, mcts_simulations, mcts_tree, temperature, model):
        a = orig(policy, board, mcts_simulations, mcts_tree, temperature, model)
        h, nn, ne = _tree_hash(mcts_tree)
        per_move["tree_hash"].append(np.frombuffer(h, dtype=np.uint8))
        per_move["n_nodes"].append(nn)
        per_move["action"].append(int(a))
        return a

    sp.select_play = wrapped
    gd = sp.play_game(net, net, sims, stop_exploration, self_play=True, num_moves=num_moves)
    moves = gd["moves"]
    data.update({
        "size": np.array(size, dtype=np.int32), "sims": np.array(sims, dtype=np.int32), "batch": np.array(batch, dtype=np.int32),
        "stop_exploration": np.array(stop_exploration, dtype=np.int32), "num_moves": np.array(num_moves, dtype=np.int32),
        "net": np.frombuffer(net_kind.encode(), dtype=np.uint8), "komi": np.array(5.5),
        "uniforms": np.array(rec["uniforms"], dtype=np.float64),
        "noises": np.array(rec["noises"], dtype=np.float64).reshape(-1, A),
        "move_index": np.array([mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S for mv in moves], dtype=np.int32),
        "move_player": np.array([mv["player"] for mv in moves], dtype=np.int8),
        "move_value": np.array([np.float32(np.asarray(mv["value"]).reshape(-1)[0]) for mv in moves], dtype=np.float32),
        "move_policy": np.array([mv["policy"] for mv in moves], dtype=np.float64).reshape(-1, A),
        "move_board_hash": np.array([_sha8(mv["board"]) for mv in moves], dtype=np.uint8).reshape(-1, 8),
        "result": np.frombuffer(gd["result"].encode(), dtype=np.uint8),
        "winner": np.array(-99 if gd["winner"] is None else gd["winner"], dtype=np.int32),
    })
    for k, v in per_move.items():
        data["pm_" + k] = np.array(v)
    np.savez_compressed(out, **data)
    print("sync S=%d sims=%d batch=%d net=%s: %d moves, result %s" % (size, sims, batch, net_kind, len(moves), gd["result"]))



# ----------------------------------------------------------------------------------------------
# the reference's own unit tests as data: every call they make into the path, with its inputs and the value
# the reference returned, recorded while the reference's test classes run (and pass) in this container
# ----------------------------------------------------------------------------------------------
_TYPE_CODES = {int: 0, float: 1}


def _tcode(v):
    import numpy as np
    if isinstance(v, (bool, int)) and not isinstance(v, np.generic):
        return 0
    if isinstance(v, float) and not isinstance(v, np.generic):
        return 1
    if isinstance(v, np.float32):
        return 2
    if isinstance(v, np.float64):
        return 3
    if isinstance(v, np.integer):
        return 4
    raise TypeError(type(v))


def ser_tree(root):
    """Pre-order rows of a dict tree (children in dict order).  ints: depth, action, present-mask, index, count,
    virtual_loss, has_parent, type codes of value / mean_value / p; floats: value, mean_value, p.
    present-mask bits: 1 index, 2 count, 4 virtual_loss, 8 value, 16 mean_value, 32 p, 64 parent key, 128 subtree is None."""
    import numpy as np
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
                     _tcode(node.get("value", 0)), _tcode(node.get("mean_value", 0)), _tcode(node.get("p", 0))])
        flts.append([float(node.get("value", 0)), float(node.get("mean_value", 0)), float(node.get("p", 0))])
        for a, c in (node.get("subtree") or {}).items():
            row(c, depth + 1, int(a))

    row(root, 0, -1)
    return np.array(ints, dtype=np.int64), np.array(flts, dtype=np.float64)


def child_units(out):
    import io
    import types
    import unittest
    import numpy as np
    _setup_reference(9, 8, 8)
    sys.path.insert(0, os.path.join(REF, "test"))
    import tests as rt
    import tree_util_tests as tt
    calls = []
    current = {"test": ""}

    def enc(v):
        """-> (kind, array): 0 none, 1 ndarray, 2 int, 3 list of int tuples, 4 {int: int}, 5 float"""
        if v is None:
            return 0, np.zeros(0, np.int8)
        if isinstance(v, np.ndarray):
            return 1, v
        if isinstance(v, (bool, int, np.integer)):
            return 2, np.array(int(v), dtype=np.int64)
        if isinstance(v, (float, np.floating)):
            return 5, np.array(float(v), dtype=np.float64)
        if isinstance(v, dict):
            return 4, np.array(sorted((int(k), int(x)) for k, x in v.items()), dtype=np.int64).reshape(-1, 2)
        if isinstance(v, (list, tuple)):
            return 3, np.array([tuple(int(t) for t in e) for e in v], dtype=np.int64).reshape(len(v), 2 if not len(v) else -1)
        raise TypeError(type(v))

    def record(name, fn, n_args, defaults=(), sort_out=False, out_index=None):
        def wrapped(*a, **k):
            args = list(a) + [k.get(d[0], d[1]) for d in defaults[len(a) - (n_args - len(defaults)):]] if len(a) < n_args else list(a)
            ins = [enc(np.copy(x) if isinstance(x, np.ndarray) else x) for x in args]
            res = fn(*a, **k)
            o = res if out_index is None else res[out_index]
            if sort_out and o is not None:
                o = sorted(o)
            extra = enc(res[1]) if out_index is not None else None
            calls.append({"fn": name, "test": current["test"], "ins": ins, "out": enc(np.copy(o) if isinstance(o, np.ndarray) else o),
                          "out2": extra})
            return res
        return wrapped

    # play.py functions as imported into the test module
    rt.color_board = record("color_board", rt.color_board, 2)
    rt._get_points = record("_get_points", rt._get_points, 1)
    rt.capture_group = record("capture_group", rt.capture_group, 3, defaults=(("group", None),))
    rt.make_play = record("make_play", rt.make_play, 4, defaults=(("color", None),), out_index=0)
    rt.legal_moves = record("legal_moves", rt.legal_moves, 1)
    rt.get_liberties = record("get_liberties", rt.get_liberties, 4, defaults=(("color", None),), sort_out=True)
    for nm in ("left_diagonal", "right_diagonal", "vertical_axis", "horizontal_axis", "rotation_90", "rotation_180", "rotation_270",
               "reverse_left_diagonal", "reverse_right_diagonal", "reverse_vertical_axis", "reverse_horizontal_axis",
               "reverse_rotation_90", "reverse_rotation_180", "reverse_rotation_270"):
        setattr(rt, nm, record(nm, getattr(rt, nm), 1))

    # simulate (sync MCTS) with the model's traffic recorded
    sims = []
    orig_sim = rt.simulate

    class Tap(object):
        def __init__(self, model, log):
            self.model, self.log = model, log
            self.name = getattr(model, "name", "model")

        def predict_on_batch(self, X):
            p, v = self.model.predict_on_batch(X)
            self.log.append((np.array(X), np.array(p), np.array(v)))
            return p, v

    depth = {"n": 0}

    def sim_wrapped(node, board, model, mcts_batch_size, original_player):
        top = depth["n"] == 0
        depth["n"] += 1
        try:
            if not top:
                return orig_sim(node, board, model, mcts_batch_size, original_player)
            ti, tf = ser_tree(node)
            log = []
            b_in = np.copy(board)
            orig_sim(node, board, Tap(model, log), mcts_batch_size, original_player)
            oi, of = ser_tree(node)
            sims.append({"test": current["test"], "tree_in": (ti, tf), "tree_out": (oi, of), "board_in": b_in, "board_out": np.copy(board),
                         "batch": int(mcts_batch_size), "orig": int(original_player), "log": log})
        finally:
            depth["n"] -= 1

    rt.simulate = sim_wrapped
    import self_play as _sp
    _sp_sim = _sp.simulate
    _sp.simulate = lambda *a, **k: sim_wrapped(*a, **k)      # the recursion inside self_play goes through the module global
    orig_sim = _sp_sim

    # tree_util / back_propagation unit tests
    trees = []

    def tree_call(name, fn):
        def wrapped(*a):
            root = a[-1] if name == "back_propagation" else a[0]
            ti, tf = ser_tree(root)
            extra = {}
            if name == "back_propagation":
                leaf, moves = a[0]
                extra["leaf"] = ser_tree(leaf)
                extra["moves"] = np.array(moves, dtype=np.int64)
            if name == "get_node_by_moves":
                extra["moves"] = np.array(a[1], dtype=np.int64)
            try:
                res = fn(*a)
                err = 0
            except Exception:
                res, err = None, 1
            oi, of = ser_tree(root)
            rec = {"fn": name, "test": current["test"], "tree_in": (ti, tf), "tree_out": (oi, of), "err": err}
            rec.update(extra)
            if name == "find_best_leaf_virtual_loss":
                node, mv = res
                rec["moves_out"] = np.array(mv if mv is not None else [-99], dtype=np.int64)
                rec["leaf_index"] = -99 if node is None else int(node.get("index", -98))
            elif name == "get_node_by_moves" and not err:
                rec["leaf_index"] = int(res.get("index", -98))
            elif name == "tree_depth":
                rec["depth"] = int(res)
            trees.append(rec)
            if err:
                raise Exception("ERROR: Unable to get node: Invalid moves array")
            return res
        return wrapped

    for nm in ("find_best_leaf_virtual_loss", "get_node_by_moves", "back_propagation", "tree_depth"):
        setattr(tt, nm, tree_call(nm, getattr(tt, nm)))
    rt.tree_depth = tree_call("tree_depth", rt.tree_depth)

    class Result(unittest.TextTestResult):
        def startTest(self, test):
            current["test"] = test.id().split(".", 1)[1]
            super().startTest(test)

    suite = unittest.TestSuite()
    for cls in (rt.TestGoMethods, rt.TestBoardMethods, rt.TestSymmetrydTestCase, rt.MCTSTestCase, tt.TreeTestCase):
        suite.addTests(unittest.defaultTestLoader.loadTestsFromTestCase(cls))
    stream = io.StringIO()
    res = unittest.TextTestRunner(stream=stream, resultclass=Result, verbosity=0).run(suite)
    assert res.wasSuccessful(), stream.getvalue()
    data = {"n_calls": np.array(len(calls)), "n_sims": np.array(len(sims)), "n_trees": np.array(len(trees)),
            "tests_run": np.array(res.testsRun), "size": np.array(9), "komi": np.array(5.5)}

    def put(key, kv):
        kind, arr = kv
        data[key + "_k"] = np.array(kind, dtype=np.int8)
        if isinstance(arr, np.ndarray) and arr.dtype in (np.int32, np.int64) and arr.size and np.abs(arr).max() < 100:
            arr = arr.astype(np.int8)
        data[key] = arr

    # make_play is called hundreds of times (test set-up sequences): stored stacked; the other calls one by one
    mp = [c for c in calls if c["fn"] == "make_play"]
    calls = [c for c in calls if c["fn"] != "make_play"]
    data["n_calls"] = np.array(len(calls))
    data["mp_x"] = np.array([int(c["ins"][0][1]) for c in mp], dtype=np.int16)
    data["mp_y"] = np.array([int(c["ins"][1][1]) for c in mp], dtype=np.int16)
    data["mp_board_in"] = np.array([c["ins"][2][1][0] for c in mp], dtype=np.int8)
    data["mp_color"] = np.array([0 if c["ins"][3][0] == 0 else int(c["ins"][3][1]) for c in mp], dtype=np.int8)
    data["mp_board_out"] = np.array([c["out"][1][0] for c in mp], dtype=np.int8)
    data["mp_player"] = np.array([int(c["out2"][1]) for c in mp], dtype=np.int8)
    for i, c in enumerate(calls):
        p = "c%03d_" % i
        data[p + "fn"] = np.frombuffer(c["fn"].encode(), dtype=np.uint8)
        data[p + "test"] = np.frombuffer(c["test"].encode(), dtype=np.uint8)
        data[p + "nin"] = np.array(len(c["ins"]))
        for j, kv in enumerate(c["ins"]):
            put(p + "in%d" % j, kv)
        put(p + "out", c["out"])
        if c["out2"] is not None:
            put(p + "out2", c["out2"])
    for i, c in enumerate(sims):
        p = "s%02d_" % i
        data[p + "test"] = np.frombuffer(c["test"].encode(), dtype=np.uint8)
        data[p + "tin_i"], data[p + "tin_f"] = c["tree_in"]
        data[p + "tout_i"], data[p + "tout_f"] = c["tree_out"]
        data[p + "board_in"] = c["board_in"].astype(np.int8)
        data[p + "board_out"] = c["board_out"].astype(np.int8)
        data[p + "batch"] = np.array(c["batch"]); data[p + "orig"] = np.array(c["orig"])
        data[p + "n_pred"] = np.array(len(c["log"]))
        for j, (X, pp, vv) in enumerate(c["log"]):
            data[p + "X%d" % j] = np.asarray(X).astype(np.int8)
            data[p + "Xdtype%d" % j] = np.frombuffer(str(np.asarray(X).dtype).encode(), dtype=np.uint8)
            data[p + "P%d" % j] = np.asarray(pp, dtype=np.float32)
            data[p + "V%d" % j] = np.asarray(vv, dtype=np.float32)
    for i, c in enumerate(trees):
        p = "t%02d_" % i
        data[p + "fn"] = np.frombuffer(c["fn"].encode(), dtype=np.uint8)
        data[p + "test"] = np.frombuffer(c["test"].encode(), dtype=np.uint8)
        data[p + "tin_i"], data[p + "tin_f"] = c["tree_in"]
        data[p + "tout_i"], data[p + "tout_f"] = c["tree_out"]
        data[p + "err"] = np.array(c["err"])
        for k in ("moves", "moves_out"):
            if k in c:
                data[p + k] = c[k]
        for k in ("leaf_index", "depth"):
            if k in c:
                data[p + k] = np.array(c[k])
        if "leaf" in c:
            data[p + "leaf_i"], data[p + "leaf_f"] = c["leaf"]
    np.savez_compressed(out, **data)
    by = {}
    for c in calls:
        by[c["fn"]] = by.get(c["fn"], 0) + 1
    print("units: %d reference tests passed; recorded %d rule/symmetry calls %s, %d simulate calls, %d tree calls" % (
        res.testsRun, len(calls), by, len(sims), len(trees)))


GTP_SCRIPT = ["protocol_version", "boardsize 9", "komi 5.5", "clear_board", "play B E5", "genmove W", "genmove B", "play W C3",
              "genmove B", "play W pass", "genmove B", "genmove W", "play B A9", "genmove W", "genmove B", "genmove W", "play B J1",
              "genmove W"]


def child_gtp(out):
    """The reference's GTP front-end (sejonggo_nomodel.py:20-160: SejongGoEngine.play / genmove on a persistent tree, GTPEngine's
    text layer) driven by a command script with the rounding-free stub net; temperature 0 and no noise, so the session is
    deterministic.  Recorded per command: the reply, the board hash, and (after a genmove) the hash of the kept subtree."""
    import copy
    import collections
    import numpy as np
    conf = _setup_reference(9, 48, 8)
    conf["GPUs"] = [0]
    import play
    import predicting_queue_worker as pq
    import simulation_workers as sw
    import nomodel_self_play as ns
    from sejonggo_amd.stub_nets import make_stub
    net = make_stub("hash", 9)
    counters = {"predict": 0}

    def stub_predict(indicator, board, response_now=False):
        counters["predict"] += 1
        p, v = net.predict_on_batch(np.asarray(board))
        return p[0], v[0][0]

    for m in (pq, sw, ns):
        m.put_predict_request = stub_predict
        m.put_name_request = lambda ind: net.name
    pq.init_predicting_workers = lambda gpus: None
    pq.destroy_predicting_workers = lambda *a: None

    class FakePool(object):
        def apply_async(self, fn, args, error_callback=None, callback=None):
            leaf, board, moves, ind, orig, pid = args
            fn(copy.deepcopy(leaf), np.copy(board), list(moves), ind, orig, pid)

        def close(self):
            pass

        def join(self):
            pass

    class FakeQueue(object):
        def __init__(self):
            self.q = collections.deque()

        def put(self, x):
            self.q.append(x)

        def get(self):
            return self.q.popleft()

    sw.init_simulation_workers_by_gpuid = lambda gpu: None
    sw.process_pool = FakePool()
    sw.simulation_result_queue[0] = FakeQueue()
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        import sejonggo_nomodel as sn
        sn.put_predict_request = stub_predict
        eng = sn.GTPEngine()
    replies, hashes, tree_hashes, n_nodes, players = [], [], [], [], []
    for cmd in GTP_SCRIPT:
        with contextlib.redirect_stdout(io.StringIO()):
            r = eng.parse_command(cmd)
        replies.append(r)
        hashes.append(_sha8(eng.sejong_engine.board))
        t = eng.sejong_engine.mcts_tree
        if cmd.startswith("genmove") and t is not None and t["subtree"]:
            h, nn, _ = _tree_hash(t)
        else:
            h, nn = b"\0" * 16, 0 if (t is None or not t["subtree"]) else -1
        tree_hashes.append(np.frombuffer(h, dtype=np.uint8))
        n_nodes.append(nn)
        players.append(int(eng.sejong_engine.board[0, 0, 0, -1]))
    np.savez_compressed(out, size=np.array(9), sims=np.array(48), energy=np.array(8), komi=np.array(5.5),
                        script=np.frombuffer("\n".join(GTP_SCRIPT).encode(), dtype=np.uint8),
                        replies=np.frombuffer("\x1e".join(replies).encode(), dtype=np.uint8),
                        board_hash=np.array(hashes, dtype=np.uint8), tree_hash=np.array(tree_hashes, dtype=np.uint8),
                        n_nodes=np.array(n_nodes, dtype=np.int64), to_play=np.array(players, dtype=np.int8),
                        n_predict=np.array(counters["predict"]), version=np.frombuffer(eng.version().encode(), dtype=np.uint8))
    print("gtp: %d commands, replies %s, %d predicts" % (len(GTP_SCRIPT), [r.strip() for r in replies], counters["predict"]))


# ----------------------------------------------------------------------------------------------
# parent side
# ----------------------------------------------------------------------------------------------
ASYNC_CASES = [
    # (size, sims, energy, net, num_moves, stop_exploration, seed)
    (9, 50, 8, "uniform", None, 30, 1),      # BASELINE.json configs[0]: 9x9, 50 sims (48 effective), stub net
    (9, 50, 8, "dummy", None, 6, 2),
    (9, 50, 8, "hash", None, 10, 3),
    (9, 200, 8, "hash", 12, 4, 4),
    (5, 48, 8, "hash", None, 4, 5),          # tiny board: few legal moves -> exercises the "no best leaf" path
    (5, 64, 16, "dummy", None, 2, 6),
    (19, 40, 8, "hash", 6, 3, 7),
    (19, 400, 8, "hash", 2, 30, 8),          # the headline search width, two plies
    (19, 1600, 32, "hash", 3, 30, 9),        # BASELINE.json configs[4]: 1600 sims, 32-leaf rounds (conf.py:18,29), three plies
    # two-model evaluation games (evaluate_worker.py:137: BEST_SYM vs LATEST_SYM, stop_exploration = 0, one tree per player)
    (9, 48, 8, "hash+hash2", None, 0, 10),   # model1 (BEST) moves first, whole game
    (9, 48, 8, "hash+hash2", 30, 0, 11),     # model2 (LATEST) moves first
    (5, 32, 8, "hash2+hash", None, 0, 13),   # tiny board: the other tree often lacks the played move
    # sizes between the tested extremes, energies other than 8 (single model)
    (13, 40, 8, "hash", 8, 3, 14),
    (7, 36, 4, "hash", None, 5, 15),         # whole 7x7 game, 4-leaf rounds
    (13, 33, 16, "uniform", 5, 5, 16),       # sims not divisible by the energy (32 effective), 16-leaf rounds
]


SYNC_CASES = [
    # (size, sims, batch, net, num_moves, stop_exploration, seed)
    (9, 48, 8, "hash", 10, 4, 11),
    (9, 32, 4, "dummy", 8, 2, 12),
    (5, 24, 8, "hash", 12, 3, 13),
]


def run_child(args, cwd):
    env = dict(os.environ)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--child"] + [str(a) for a in args]
    print("+", " ".join(cmd[2:]), flush=True)
    subprocess.check_call(cmd, cwd=cwd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", nargs="+")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    if a.child:
        what = a.child[0]
        if what == "rules":
            child_rules(int(a.child[1]), a.child[2])
        elif what == "sgf":
            child_sgf(a.child[1])
        elif what == "sym":
            child_sym(int(a.child[1]), a.child[2])
        elif what == "puct":
            child_puct(a.child[1])
        elif what == "units":
            child_units(a.child[1])
        elif what == "gtp":
            child_gtp(a.child[1])
        elif what == "sync":
            s_, sims, b, net, nm, se, seed, out = a.child[1:]
            child_sync(int(s_), int(sims), int(b), net, int(nm), int(se), int(seed), out)
        elif what == "async":
            s, sims, e, net, nm, se, seed, out = a.child[1:]
            child_async(int(s), int(sims), int(e), net, None if nm == "None" else int(nm), int(se), int(seed), out)
        return
    scratch = tempfile.mkdtemp(prefix="sgo_golden_")
    only = a.only
    if only in (None, "rules"):
        for s in (5, 7, 9, 13, 19):
            run_child(["rules", s, os.path.join(HERE, "rules_S%d.npz" % s)], scratch)
    if only in (None, "sgf"):
        run_child(["sgf", os.path.join(HERE, "sgf_S19.npz")], scratch)
    if only in (None, "sym"):
        for s in (5, 9, 19):
            run_child(["sym", s, os.path.join(HERE, "sym_S%d.npz" % s)], scratch)
    if only in (None, "puct"):
        run_child(["puct", os.path.join(HERE, "puct.npz")], scratch)
    if only in (None, "units"):
        run_child(["units", os.path.join(HERE, "units_S9.npz")], scratch)
    if only in (None, "gtp"):
        run_child(["gtp", os.path.join(HERE, "gtp_S9.npz")], scratch)
    if only in (None, "sync"):
        for i, c in enumerate(SYNC_CASES):
            run_child(["sync"] + list(c) + [os.path.join(HERE, "sync_%02d.npz" % i)], scratch)
    if only is not None and only.startswith("async_"):
        i = int(only.split("_")[1])
        run_child(["async"] + list(ASYNC_CASES[i]) + [os.path.join(HERE, "async_%02d.npz" % i)], scratch)
    if only in (None, "async"):
        for i, c in enumerate(ASYNC_CASES):
            run_child(["async"] + list(c) + [os.path.join(HERE, "async_%02d.npz" % i)], scratch)


if __name__ == "__main__":
    main()
