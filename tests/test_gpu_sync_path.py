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
