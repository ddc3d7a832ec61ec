"""GPU tests of the sync self-play path mirror (sejonggo_amd.self_play: simulate / mcts_decision / play_game) against
golden outputs of the reference's self_play.py (tests/golden/sync_*.npz).  Trees are host dicts; rules + symmetry
calls go through libsgo_hip.so."""
import numpy as np
import pytest

from tests.helpers import load, sha8, dict_tree_hash, SYNC_FILES

pytestmark = pytest.mark.gpu


@pytest.fixture()
def sync_env():
    from sejonggo_amd import _lib, symmetry
    from sejonggo_amd.conf import conf
    _lib.require_gpu()
    keep, keep_sym = dict(conf), list(symmetry.SYMMETRIES)
    symmetry.SYMMETRIES[:] = symmetry.SYMMETRIES[0:1]     # identity only, like the reference's MCTSTestCase.setUp
    yield conf
    symmetry.SYMMETRIES[:] = keep_sym
    conf.clear()
    conf.update(keep)


@pytest.mark.parametrize("fn", SYNC_FILES)
def test_simulate_matches_reference(sync_env, fn):
    from sejonggo_amd import play, self_play
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S, batch = int(z["size"]), int(z["batch"])
    sync_env.update({'SIZE': S, 'MCTS_BATCH_SIZE': batch})
    net = make_stub(bytes(z["net"]).decode(), S)
    board = z["sim_board"].astype(np.int32).reshape(1, S, S, 17)
    pol, _ = net.predict_on_batch(board)
    tree = play.new_tree(pol[0], board, add_noise=False)
    for it in range(len(z["sim_hashes"])):
        self_play.simulate(tree, np.copy(board), net, batch, board[0, 0, 0, -1])
        h, nn = dict_tree_hash(tree)
        assert nn == z["sim_counts"][it][0] and tree['count'] == z["sim_counts"][it][2], it
        assert h == z["sim_hashes"][it].tobytes(), it
    assert np.float32(tree['value']).tobytes() == z["sim_root_value"].tobytes()


@pytest.mark.parametrize("fn", SYNC_FILES)
def test_sync_play_game_matches_reference(sync_env, fn, monkeypatch):
    from sejonggo_amd import self_play
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S = int(z["size"])
    sync_env.update({'SIZE': S, 'MCTS_BATCH_SIZE': int(z["batch"]), 'KOMI': float(z["komi"])})
    net = make_stub(bytes(z["net"]).decode(), S)
    uni, noi = list(z["uniforms"]), list(z["noises"])

    def fake_choice(moves, size=1, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return [moves[int(np.searchsorted(cdf, uni.pop(0), side="right"))]]

    monkeypatch.setattr(np.random, "choice", fake_choice)
    monkeypatch.setattr(np.random, "dirichlet", lambda alpha: noi.pop(0))
    hashes = []
    orig = self_play.select_play

    def wrapped(policy, board, sims, tree, temperature, model):
        a = orig(policy, board, sims, tree, temperature, model)
        hashes.append(dict_tree_hash(tree)[0])
        return a

    monkeypatch.setattr(self_play, "select_play", wrapped)
    gd = self_play.play_game(net, net, int(z["sims"]), int(z["stop_exploration"]), self_play=True, num_moves=int(z["num_moves"]))
    assert len(gd['moves']) == len(z["move_index"])
    for i, mv in enumerate(gd['moves']):
        a = mv['move'][0] + S * mv['move'][1] if mv['move'][1] != S else S * S
        assert a == z["move_index"][i] and mv['player'] == z["move_player"][i], i
        assert np.float32(np.asarray(mv['value']).reshape(-1)[0]).tobytes() == z["move_value"][i].tobytes(), i
        assert mv['policy'].tobytes() == z["move_policy"][i].tobytes(), i
        assert np.array_equal(sha8(mv['board']), z["move_board_hash"][i]), i
        assert hashes[i] == z["pm_tree_hash"][i].tobytes(), i
    assert gd['result'] == bytes(z["result"]).decode()


@pytest.mark.parametrize("fn", ["async_02.npz", "async_05.npz"])
def test_host_async_path_matches_reference(sync_env, fn, monkeypatch):
    """nomodel_self_play.play_game_host (host dict trees, async_simulate2 + back_propagation mirrors) replays the
    reference's golden async games, including the game that hits the 'No best leaf' path 24 times."""
    from sejonggo_amd import nomodel_self_play as ns, predicting_queue_worker as pq
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import dict_tree_hash
    z = load(fn)
    S = int(z["size"])
    nm = int(z["num_moves"])
    sync_env.update({'SIZE': S, 'MCTS_SIMULATIONS': int(z["sims"]), 'ENERGY': int(z["energy"]), 'KOMI': float(z["komi"])})
    net = make_stub(bytes(z["net"]).decode(), S)
    pq.set_model_factory(lambda kind: net)
    pq.init_predicting_workers([0])
    uni, noi = list(z["uniforms"]), list(z["noises"])

    def fake_choice(moves, size=1, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return [moves[int(np.searchsorted(cdf, uni.pop(0), side="right"))]]

    monkeypatch.setattr(np.random, "choice", fake_choice)
    monkeypatch.setattr(np.random, "dirichlet", lambda alpha: noi.pop(0))
    hashes = []
    orig = ns.select_play

    def wrapped(board, energy, tree, temperature, indicator, gpuid):
        a = orig(board, energy, tree, temperature, indicator, gpuid)
        hashes.append(dict_tree_hash(tree)[0])
        return a

    monkeypatch.setattr(ns, "select_play", wrapped)
    try:
        gd = ns.play_game_host("BEST_SYM", "BEST_SYM", int(z["energy"]), int(z["stop_exploration"]), 0, self_play=True,
                               num_moves=None if nm < 0 else nm)
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
    assert len(gd['moves']) == len(z["move_index"])
    for i, mv in enumerate(gd['moves']):
        a = mv['move'][0] + S * mv['move'][1] if mv['move'][1] != S else S * S
        assert a == z["move_index"][i] and mv['player'] == z["move_player"][i], i
        assert mv['policy'].tobytes() == z["move_policy"][i].tobytes(), i
        assert hashes[i] == z["pm_tree_hash"][i].tobytes(), i
    assert gd['result'] == bytes(z["result"]).decode()


def test_gtp_front_end(sync_env):
    """GTP text loop (sejonggo_nomodel.py:76-185): vertex mapping skips 'I' and counts rows from the bottom; genmove
    plays a legal move on the engine's board and keeps the searched subtree."""
    import io
    from sejonggo_amd import gtp, predicting_queue_worker as pq
    from sejonggo_amd.play import legal_moves
    from sejonggo_amd.stub_nets import make_stub
    sync_env.update({'SIZE': 9, 'MCTS_SIMULATIONS': 32, 'ENERGY': 8, 'GPUs': [0]})
    net = make_stub("hash", 9)
    pq.set_model_factory(lambda kind: net)
    try:
        e = gtp.GTPEngine()
        assert e.parse_move("A9") == (0, 0) and e.parse_move("J1") == (8, 8) and e.parse_move("pass") == (0, 9)
        assert e.print_move(0, 0) == "A9" and e.print_move(8, 8) == "J1" and e.print_move(7, 4) == "H5"
        assert e.parse_command("protocol_version") == "= 2\n\n" and e.parse_command("boardsize 9") == "=\n\n"
        assert e.parse_command("play B E5") == "=\n\n" and e.board[0, 4, 4, 1] == 1     # black stone, white to play
        before = e.board.copy()
        mask = legal_moves(before)
        reply = e.parse_command("genmove W")
        assert reply.startswith("= ") and reply.endswith("\n\n")
        x, y = e.parse_move(reply[2:].strip())
        a = 81 if y == 9 else y * 9 + x
        assert mask[a] == 0 and e.sejong_engine.move == 3
        assert "name" in e.parse_command("list_commands") and e.parse_command("bogus").startswith("?")
        out = io.StringIO()
        gtp.main(io.StringIO("clear_board\ngenmove B\nquit\n"), out)
        assert out.getvalue().count("=") == 3
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])


TWO_MODEL_FILES = ["async_09.npz", "async_10.npz", "async_11.npz"]


@pytest.mark.parametrize("fn", TWO_MODEL_FILES)
def test_two_model_evaluation_game_matches_reference(sync_env, fn, monkeypatch):
    """The evaluator's game (evaluate_worker.py:137): play_game_async("BEST_SYM", "LATEST_SYM", energy, stop_exploration=0)
    with two DIFFERENT nets, one tree per player, the mover's tree re-rooted and the other tree followed when it holds
    the move (nomodel_self_play.py:203-218).  Fixtures recorded from the reference with either model moving first;
    compared move for move, tree for tree (both players' trees, after every search), and in the game_data fields the
    evaluator reads -- including winner_model, which the reference gets wrong when model1 plays white (:247)."""
    from sejonggo_amd import nomodel_self_play as ns, play, predicting_queue_worker as pq
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import dict_tree_hash, name_of
    z = load(fn)
    assert int(z["two_model"]) == 1
    S = int(z["size"])
    nm = int(z["num_moves"])
    sync_env.update({'SIZE': S, 'MCTS_SIMULATIONS': int(z["sims"]), 'ENERGY': int(z["energy"]), 'KOMI': float(z["komi"]),
                     'COMPAT_LATEST_SYM': False, 'COMPAT_WINNER_MODEL': True})
    kinds = name_of(z, "net").split("+")
    nets = {"BEST": make_stub(kinds[0], S), "LATEST": make_stub(kinds[1], S)}
    seen = {"BEST": 0, "LATEST": 0}

    class Counting(object):
        def __init__(self, kind):
            self.kind, self.name, self.numpy_native = kind, nets[kind].name, True

        def predict_on_batch(self, X):
            seen[self.kind] += len(X)
            return nets[self.kind].predict_on_batch(X)

    wrapped_nets = {k: Counting(k) for k in nets}
    pq.set_model_factory(lambda kind: wrapped_nets[kind])
    pq.init_predicting_workers([0])
    monkeypatch.setattr(play, "random", lambda: float(z["first_draw"]))
    trace = []
    orig = ns.select_play

    def wrapped(board, energy, tree, temperature, indicator, gpuid):
        a = orig(board, energy, tree, temperature, indicator, gpuid)
        trace.append((dict_tree_hash(tree)[0], 0 if indicator.startswith("BEST") else 1, temperature, tree['count']))
        return a

    monkeypatch.setattr(ns, "select_play", wrapped)
    try:
        gd = ns.play_game_async("BEST_SYM", "LATEST_SYM", int(z["energy"]), int(z["stop_exploration"]), 0,
                                num_moves=None if nm < 0 else nm)
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
    assert len(gd['moves']) == len(z["move_index"])
    for i, mv in enumerate(gd['moves']):
        a = mv['move'][0] + S * mv['move'][1] if mv['move'][1] != S else S * S
        assert a == z["move_index"][i] and mv['player'] == z["move_player"][i], i
        assert np.float32(mv['value']).tobytes() == z["move_value"][i].tobytes(), i
        assert mv['policy'].tobytes() == z["move_policy"][i].tobytes(), i
        assert np.array_equal(sha8(mv['board']), z["move_board_hash"][i]), i
        h, model, temp, root_count = trace[i]
        assert h == z["pm_tree_hash"][i].tobytes() and model == z["pm_model"][i], i
        assert temp == z["pm_temperature"][i] == 0 and root_count == z["pm_root_count"][i], i
    assert gd['result'] == name_of(z, "result")
    assert (-99 if gd['winner'] is None else gd['winner']) == int(z["winner"])
    assert gd['modelB_name'] == name_of(z, "modelB_name") and gd['modelW_name'] == name_of(z, "modelW_name")
    assert (gd['winner_model'] or "") == name_of(z, "winner_model")
    assert seen["BEST"] == int(z["n_predict_best"]) and seen["LATEST"] == int(z["n_predict_latest"])
    # the quirk, made explicit: with model1 (BEST) on white the reference names the loser
    model1_black = float(z["first_draw"]) < .5
    if int(z["winner"]) in (0, 1):
        true_winner = gd['modelB_name'] if int(z["winner"]) == 1 else gd['modelW_name']
        assert (gd['winner_model'] == true_winner) == model1_black


def test_gtp_session_matches_reference(sync_env):
    """The reference's GTP front-end (sejonggo_nomodel.py:20-160) replayed command for command: replies, the board after
    every command and the kept subtree after every genmove (tests/golden/gtp_S9.npz, recorded from the reference with the
    rounding-free stub net; temperature 0, no noise -> deterministic)."""
    from sejonggo_amd import gtp, predicting_queue_worker as pq
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import dict_tree_hash, name_of
    z = load("gtp_S9.npz")
    S = int(z["size"])
    sync_env.update({'SIZE': S, 'MCTS_SIMULATIONS': int(z["sims"]), 'ENERGY': int(z["energy"]), 'KOMI': float(z["komi"]), 'GPUs': [0]})
    net = make_stub("hash", S)
    seen = [0]

    class Counting(object):
        name, numpy_native = net.name, True

        def predict_on_batch(self, X):
            seen[0] += len(X)
            return net.predict_on_batch(X)

    pq.set_model_factory(lambda kind: Counting())
    try:
        e = gtp.GTPEngine()
        script = name_of(z, "script").split("\n")
        replies = name_of(z, "replies").split("\x1e")
        assert len(script) == len(replies) == len(z["board_hash"])
        for i, cmd in enumerate(script):
            assert e.parse_command(cmd) == replies[i], (i, cmd)
            assert np.array_equal(sha8(e.sejong_engine.board), z["board_hash"][i]), (i, cmd)
            assert e.sejong_engine.board[0, 0, 0, -1] == z["to_play"][i]
            t = e.sejong_engine.mcts_tree
            if cmd.startswith("genmove") and int(z["n_nodes"][i]) > 0:
                h, nn = dict_tree_hash(t)
                assert nn == int(z["n_nodes"][i]) and h == z["tree_hash"][i].tobytes(), (i, cmd)
            elif int(z["n_nodes"][i]) == 0:
                assert t is None or not t['subtree'], (i, cmd)
        assert seen[0] == int(z["n_predict"])
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])


@pytest.mark.parametrize("seed", [201, 202, 203, 204, 205, 206])
def test_fuzzed_two_model_games_device_equals_host(sync_env, seed, monkeypatch):
    """Two-model games inside k_search against the host dict-tree game loop (itself pinned by the reference's two-model
    goldens) on randomly drawn configurations: board size, energy, simulations, exploration cut-off with injected draws,
    either model moving first, a resign threshold per model.  Moves, values, policy targets, result and per-model
    evaluation counts must agree exactly."""
    from sejonggo_amd import nomodel_self_play as ns, play, predicting_queue_worker as pq
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    rng = np.random.RandomState(seed)
    S = int(rng.choice([5, 7, 9]))
    E = int(rng.choice([1, 2, 4, 8, 16]))
    sims = E * int(rng.randint(2, 6)) + int(rng.randint(0, E))
    nm = int(rng.randint(4, {5: 30, 7: 18, 9: 12}[S]))
    stop = int(rng.choice([0, 0, rng.randint(1, nm + 1)]))
    first_draw = float(rng.rand())
    r1 = None if rng.rand() < 0.4 else float(rng.uniform(-1, 0.5))
    r2 = None if rng.rand() < 0.4 else float(rng.uniform(-1, 0.5))
    uni = rng.random_sample(nm)
    nets = {"BEST": make_stub("hash", S), "LATEST": make_stub("hash2", S)}
    cfg = (S, E, sims, nm, stop, first_draw, r1, r2)
    # device
    eng = SelfPlayEngine(nets["BEST"], net2=nets["LATEST"], size=S, n_games=1, sims=sims, energy=E, stop_exploration=stop,
                         num_moves=nm, komi=5.5, symmetry="identity")
    eng.start_eval_games([0], first_model=[0 if first_draw < .5 else 1], uniforms=uni[None, :], resign_model1=r1, resign_model2=r2)
    dev = eng.run()
    n_dev = list(eng.n_model_positions)
    eng.close()
    # host
    sync_env.update({'SIZE': S, 'MCTS_SIMULATIONS': sims, 'ENERGY': E, 'KOMI': 5.5, 'COMPAT_LATEST_SYM': False,
                     'COMPAT_WINNER_MODEL': True})
    seen = {"BEST": 0, "LATEST": 0}

    class Counting(object):
        def __init__(self, kind):
            self.kind, self.name, self.numpy_native = kind, nets[kind].name, True

        def predict_on_batch(self, X):
            seen[self.kind] += len(X)
            return nets[self.kind].predict_on_batch(X)

    wrapped = {k: Counting(k) for k in nets}
    pq.set_model_factory(lambda kind: wrapped[kind])
    pq.init_predicting_workers([0])
    draws = list(uni)

    def fake_choice(moves, size=1, p=None):
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        return [moves[int(np.searchsorted(cdf, draws.pop(0), side="right"))]]

    monkeypatch.setattr(np.random, "choice", fake_choice)
    monkeypatch.setattr(play, "random", lambda: first_draw)
    try:
        gd = ns.play_game_async("BEST_SYM", "LATEST_SYM", E, stop, 0, num_moves=nm, resign_model1=r1, resign_model2=r2)
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
    d = dev[0] if dev else {"moves": [], "result": None}
    assert len(d["moves"]) == len(gd["moves"]), cfg
    for i, (a, b) in enumerate(zip(d["moves"], gd["moves"])):
        assert tuple(a["move"]) == tuple(b["move"]) and a["player"] == b["player"], (cfg, i)
        assert np.float32(a["value"]).tobytes() == np.float32(b["value"]).tobytes(), (cfg, i)
        assert np.asarray(a["policy"]).tobytes() == np.asarray(b["policy"]).tobytes(), (cfg, i)
    if gd["moves"]:
        assert d["result"] == gd["result"] and d["winner"] == gd["winner"], cfg
        assert d["modelB_name"] == gd["modelB_name"] and d["modelW_name"] == gd["modelW_name"], cfg
        assert (d["winner_model"] or "") == (gd["winner_model"] or ""), cfg
    assert n_dev == [seen["BEST"], seen["LATEST"]], cfg
