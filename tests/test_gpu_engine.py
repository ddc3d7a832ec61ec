"""GPU parity tests of the self-play engine (k_search / k_compact / board_advance through sgo_step):
every golden game recorded from the Python reference must be reproduced move for move, with bit-identical
root tables, whole-tree hashes, policy targets and results; many concurrent games must each equal the
oracle's game; symmetry handling (fixed k, 8-fold average) must equal the oracle driven through the same
transforms."""
import hashlib

import numpy as np
import pytest

from tests.helpers import load, sha8, ASYNC_FILES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from sejonggo_amd import _lib
    _lib.require_gpu()
    return _lib


def _engine(z, net, halt_at=None, **kw):
    from sejonggo_amd.engine import SelfPlayEngine
    S = int(z["size"])
    nm = int(z["num_moves"])
    eng = SelfPlayEngine(net, size=S, n_games=1, sims=int(z["sims"]), energy=int(z["energy"]),
                         stop_exploration=int(z["stop_exploration"]), num_moves=None if nm < 0 else nm,
                         komi=float(z["komi"]), symmetry="identity", **kw)
    uni = np.zeros((1, max(1, eng.max_moves)))
    uni[0, :len(z["uniforms"])] = z["uniforms"]
    eng.start_games([0], noises=z["noises"][:1], uniforms=uni)
    if halt_at is not None:
        eng.set_halt(0, halt_at)
    return eng


@pytest.mark.parametrize("fn", ASYNC_FILES)
def test_golden_games(L, fn):
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S = int(z["size"])
    net = make_stub(bytes(z["net"]).decode(), S)
    eng = _engine(z, net)
    games = eng.run()
    assert len(games) == 1
    gd = games[0]
    n_moves = len(z["move_index"])
    assert len(gd["moves"]) == n_moves
    for i, mv in enumerate(gd["moves"]):
        a = mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S
        assert a == z["move_index"][i], i
        assert tuple(mv["move"]) == tuple(z["move_xy"][i]), i
        assert mv["player"] == z["move_player"][i], i
        assert mv["move_n"] == i
        assert mv["value"].tobytes() == z["move_value"][i].tobytes(), i
        assert mv["board"].dtype == np.int32 and np.array_equal(sha8(mv["board"]), z["move_board_hash"][i]), i
        assert mv["policy"].dtype == np.float64 and mv["policy"].tobytes() == z["move_policy"][i].tobytes(), i
    assert gd["result"] == bytes(z["result"]).decode()
    assert (-99 if gd["winner"] is None else gd["winner"]) == int(z["winner"])
    st = eng.status
    assert st.total_evals == int(z["n_predict"])
    assert st.none_events == int(z["none_events"])
    assert st.total_moves == n_moves
    eng.close()


@pytest.mark.parametrize("fn", ASYNC_FILES)
def test_golden_trees(L, fn):
    """Root child tables and the canonical whole-tree hash right after the search of selected moves."""
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S = int(z["size"])
    net = make_stub(bytes(z["net"]).decode(), S)
    n_moves = len(z["move_index"])
    for k in sorted(set([0, 1, min(3, n_moves - 1), n_moves // 2, n_moves - 1])):
        eng = _engine(z, net, halt_at=k)
        eng.run()
        t = eng.root_table(0)
        assert np.array_equal(t["N"], z["pm_N"][k]), k
        assert t["W"].tobytes() == z["pm_W"][k].tobytes(), k
        assert t["Q"].tobytes() == z["pm_Q"][k].tobytes(), k
        assert t["P"].tobytes() == z["pm_P"][k].tobytes(), k
        assert np.array_equal(t["EX"], z["pm_EX"][k]), k
        assert t["root_count"] == z["pm_root_count"][k], k
        assert t["root_value"].tobytes() == z["pm_root_value"][k].tobytes(), k
        buf, nn, ne = eng.tree_serialize(0)
        assert nn == z["pm_n_nodes"][k] and ne == z["pm_n_expanded"][k], k
        assert hashlib.sha1(buf.tobytes()).digest()[:16] == z["pm_tree_hash"][k].tobytes(), k
        eng.close()


def test_many_concurrent_games_equal_the_oracle(L):
    """64 games with different draws share one GPU context; each must equal the oracle's game."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 9, 48, 8, 64, 14
    net = make_stub("hash", S)
    rng = np.random.RandomState(11)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=5, num_moves=nm, komi=5.5,
                         symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = eng.run()
    assert len(games) == G
    for gd in games:
        s = gd["slot"]
        g = ora.Game(S, sims, E, 5, nm, uniforms=uni[s], noises=noises[s:s + 1]).run(net)
        assert g.n_moves == len(gd["moves"])
        for i, mv in enumerate(gd["moves"]):
            m = g.move(i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes(), (s, i)
            assert mv["player"] == m["player"] and mv["value"].tobytes() == m["value"].tobytes()
        r = g.result()
        assert {1: 1, -1: 0, 0: None}[r["winner"]] == gd["winner"]
        ta, _, _ = eng.tree_serialize(s)
        tb, _, _ = g.tree_serialize()
        assert ta.tobytes() == tb.tobytes(), s
    eng.close()


def _compare_with_oracle(eng, games, slots, S, sims, E, stop, nm, uni, noises, net, komi=5.5):
    from oracle import oracle as ora
    by_slot = {gd["slot"]: gd for gd in games}
    for s in slots:
        g = ora.Game(S, sims, E, stop, nm, komi=komi, uniforms=uni[s], noises=noises[s:s + 1]).run(net)
        gd = by_slot[s]
        assert g.n_moves == len(gd["moves"]) == nm, s
        for i, mv in enumerate(gd["moves"]):
            m = g.move(i)
            assert mv["move"][0] + S * mv["move"][1] == m["action"] or (mv["move"][1] == S and m["action"] == S * S), (s, i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes(), (s, i)
            assert mv["player"] == m["player"] and mv["value"].tobytes() == m["value"].tobytes(), (s, i)
        ta, na, _ = eng.tree_serialize(s)
        tb, nb, _ = g.tree_serialize()
        assert na == nb and ta.tobytes() == tb.tobytes(), s
        t = eng.root_table(s)
        o = g.root_table()
        for k in ("N", "W", "Q", "P"):
            assert t[k].tobytes() == np.asarray(o[k]).astype(t[k].dtype).tobytes(), (s, k)


def test_full_length_19x19_games_equal_the_oracle(L):
    """Six 19x19 games played to their END (2 * 361 plies, double pass or resignation): hundreds of re-roots with block
    recycling, captures, ko shapes, passes, the temperature switch at move 30 and area scoring, against the oracle move for
    move.  A thin search (16 sims) keeps the oracle's side to seconds; the rules / tree bookkeeping per ply is what is covered."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G = 19, 16, 8, 6
    net = make_stub("hash", S)
    rng = np.random.RandomState(99)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, 2 * S * S))
    resign = [None, None, None, -0.85, None, -0.9]
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, komi=5.5, symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni, resign=resign)
    games = {gd["slot"]: gd for gd in eng.run()}
    res = eng.results()
    assert len(games) == G
    lengths = []
    for s in range(G):
        g = ora.Game(S, sims, E, 30, None, uniforms=uni[s], noises=noises[s:s + 1], resign=resign[s]).run(net)
        r = g.result()
        assert g.n_moves == len(games[s]["moves"]) == res[s]["n_moves"], s
        assert r["end_reason"] == res[s]["end_reason"] and r["winner"] == res[s]["winner"], s
        assert r["black"] == res[s]["black"] and r["white"] == res[s]["white"] and r["last_player"] == res[s]["last_player"], s
        for i, mv in enumerate(games[s]["moves"]):
            m = g.move(i)
            assert mv["action"] == m["action"] and mv["player"] == m["player"], (s, i)
            assert mv["value"].tobytes() == m["value"].tobytes() and mv["policy"].tobytes() == m["policy"].tobytes(), (s, i)
            if i % 37 == 0 or i == g.n_moves - 1:
                assert np.array_equal(mv["board"], m["board"]), (s, i)
        lengths.append(g.n_moves)
    assert max(lengths) > 150                      # real full-length games, not early stops
    eng.close()


def test_config5_search_width_equals_the_oracle(L):
    """BASELINE config 5's workload -- 19x19, 1 600 sims per move in rounds of 32 leaves (conf.py:18,29) -- with the
    default block pool (20 * sims + 128 = 32 128 blocks per game, k_search's > 64 KiB dynamic-LDS path): two concurrent
    games, two plies, every move, policy target, whole tree and root table equal to the oracle's."""
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 19, 1600, 32, 2, 2
    net = make_stub("hash", S)
    rng = np.random.RandomState(55)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, num_moves=nm, komi=5.5,
                         symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = eng.run()
    assert len(games) == G and eng.status.total_evals == G * nm * (1 + sims)
    _compare_with_oracle(eng, games, range(G), S, sims, E, 30, nm, uni, noises, net)
    eng.close()


def test_headline_batch_shape_equals_the_oracle_on_sampled_slots(L):
    """The bench's own shape: 1 024 concurrent 19x19 games, 400 sims per move, 8 leaves per game and round = 8 192-leaf
    launches of board_advance and 8 192-position evaluations.  Two plies; a handful of slots spread over the batch are
    checked move for move and tree for tree against the oracle (the rest share the kernels and the launch)."""
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 19, 400, 8, 1024, 2
    net = make_stub("hash", S)
    rng = np.random.RandomState(77)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, num_moves=nm, komi=5.5,
                         symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = eng.run()
    assert len(games) == G and eng.status.total_moves == G * nm
    assert eng.status.total_evals == G * nm * (1 + sims) and eng.status.none_events == 0
    ms, launches, leaves = eng.advance_timing()
    assert leaves == G * nm * sims and launches == nm * (sims // E)          # every launch carried all 8 192 leaves
    _compare_with_oracle(eng, games, [0, 1, 63, 64, 511, 777, 1023], S, sims, E, 30, nm, uni, noises, net)
    eng.close()


def test_headline_batch_shape_from_the_shared_pool(L):
    """1 024 concurrent 19x19 games whose private regions hold 10 blocks each: every round 8 192 overflow ids are backed by
    8 192 concurrent atomic pops of ONE shared free stack (one per wavefront), every move's re-root returns thousands of blocks
    through the return list, k_compact merges them.  Two plies; sampled slots against the oracle, pool accounting exact."""
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 19, 400, 8, 1024, 2
    net = make_stub("hash", S)
    rng = np.random.RandomState(79)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    pool = G * 900
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, num_moves=nm, komi=5.5,
                         symmetry="identity", blocks_per_game=10, shared_blocks=pool)
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = eng.run()
    assert len(games) == G and eng.status.total_moves == G * nm and eng.status.none_events == 0
    assert eng.status.total_evals == G * nm * (1 + sims)
    _compare_with_oracle(eng, games, [0, 1, 63, 64, 500, 777, 1023], S, sims, E, 30, nm, uni, noises, net)
    info = eng.pool_info()
    res = eng.results()
    held = sum(max(0, int(r["blocks_high_water"]) - 10) for r in res)         # an upper bound of what the games still hold
    assert info["shared_blocks"] == pool and 0 < info["shared_free_low_water"] <= info["shared_free"] <= pool
    assert pool - info["shared_free"] <= held                                 # nothing leaked: blocks out of the pool are in trees
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)                # restarts hand everything back
    eng.step()
    assert eng.pool_info()["shared_free"] == pool
    eng.close()


TWO_MODEL_FILES = ["async_09.npz", "async_10.npz", "async_11.npz"]


def _eval_engine(z, halt_at=None, **kw):
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import name_of
    S, nm = int(z["size"]), int(z["num_moves"])
    kinds = name_of(z, "net").split("+")
    wrap = kw.pop("wrap", lambda net, size: net)
    eng = SelfPlayEngine(wrap(make_stub(kinds[0], S), S), net2=wrap(make_stub(kinds[1], S), S), size=S, n_games=1, sims=int(z["sims"]),
                         energy=int(z["energy"]), stop_exploration=int(z["stop_exploration"]), num_moves=None if nm < 0 else nm,
                         komi=float(z["komi"]), symmetry="identity", **kw)
    eng.start_eval_games([0], first_model=[0 if float(z["first_draw"]) < .5 else 1])
    if halt_at is not None:
        eng.set_halt(0, halt_at)
    return eng


@pytest.mark.parametrize("pool", [{}, {"blocks_per_game": 12, "shared_blocks": 4000}], ids=["private_blocks", "shared_pool"])
@pytest.mark.parametrize("fn", TWO_MODEL_FILES)
def test_two_model_games_on_the_device_equal_the_reference(L, fn, pool):
    """Evaluation games (evaluate_worker.py:137) inside k_search: two root pointers per slot, the tree of the side not to
    move follows the move when it holds it, each evaluation row tagged with the model that is to move.  Against the
    reference's two-model goldens: every move, value, policy target and position, the searching player's whole tree after
    selected moves, evaluations per model, result, colours and winner_model (incl. the reference's slip when model1 is white)."""
    from tests.helpers import name_of
    z = load(fn)
    S = int(z["size"])
    eng = _eval_engine(z, **pool)
    games = eng.run()
    assert len(games) == 1
    gd = games[0]
    n_moves = len(z["move_index"])
    assert len(gd["moves"]) == n_moves
    for i, mv in enumerate(gd["moves"]):
        a = mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S
        assert a == z["move_index"][i] and mv["player"] == z["move_player"][i], i
        assert mv["value"].tobytes() == z["move_value"][i].tobytes(), i
        assert mv["policy"].tobytes() == z["move_policy"][i].tobytes(), i
        assert np.array_equal(sha8(mv["board"]), z["move_board_hash"][i]), i
    assert gd["result"] == name_of(z, "result")
    assert (-99 if gd["winner"] is None else gd["winner"]) == int(z["winner"])
    assert gd["modelB_name"] == name_of(z, "modelB_name") and gd["modelW_name"] == name_of(z, "modelW_name")
    assert (gd["winner_model"] or "") == name_of(z, "winner_model")
    assert eng.n_model_positions == [int(z["n_predict_best"]), int(z["n_predict_latest"])]
    assert eng.status.total_evals == int(z["n_predict"]) and eng.status.none_events == int(z["none_events"])
    eng.close()
    for k in sorted(set([0, 1, 2, 3, n_moves // 2, n_moves - 2, n_moves - 1])):
        eng = _eval_engine(z, halt_at=k, **pool)
        eng.run()
        t = eng.root_table(0)
        assert np.array_equal(t["N"], z["pm_N"][k]) and t["W"].tobytes() == z["pm_W"][k].tobytes(), k
        assert t["Q"].tobytes() == z["pm_Q"][k].tobytes() and t["P"].tobytes() == z["pm_P"][k].tobytes(), k
        assert t["root_count"] == z["pm_root_count"][k] and t["root_value"].tobytes() == z["pm_root_value"][k].tobytes(), k
        buf, nn, ne = eng.tree_serialize(0)
        assert nn == z["pm_n_nodes"][k] and ne == z["pm_n_expanded"][k], k
        assert hashlib.sha1(buf.tobytes()).digest()[:16] == z["pm_tree_hash"][k].tobytes(), k
        eng.close()


@pytest.mark.parametrize("fn", TWO_MODEL_FILES)
def test_two_model_games_on_the_packed_record_route(L, fn):
    """Two-model games with BOTH nets on the packed-record route: the rows of a mixed evaluation list are split per model by
    gathering the device-side index list (engine._index_view over the engine's own memory) and each model's stem kernel reads
    its rows' records through that sub-list.  Moves, values, priors and per-model evaluation counts against the reference's
    goldens."""
    from tests.helpers import name_of
    z = load(fn)
    S = int(z["size"])
    eng = _eval_engine(z, wrap=_PackedProbe)
    assert eng.packed and eng.two_model
    games = eng.run()
    gd = games[0]
    assert len(gd["moves"]) == len(z["move_index"])
    for i, mv in enumerate(gd["moves"]):
        a = mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S
        assert a == z["move_index"][i] and mv["player"] == z["move_player"][i], i
        assert mv["value"].tobytes() == z["move_value"][i].tobytes() and mv["policy"].tobytes() == z["move_policy"][i].tobytes(), i
    assert gd["result"] == name_of(z, "result")
    assert eng.n_model_positions == [int(z["n_predict_best"]), int(z["n_predict_latest"])]
    eng.close()


@pytest.mark.parametrize("packed", [False, True], ids=["tensor_route", "packed_records"])
def test_many_two_model_games_share_a_context(L, packed):
    """32 concurrent evaluation games with both colour assignments in one context: every game equals the single-game run
    with the same first player (games do not interact), and both nets see roughly half of the positions.  On the packed
    route every mixed evaluation list is split per model by gathering the device-side index list: each net's stem kernel
    reads its rows' records through its own sub-list."""
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, nm, G = 9, 32, 8, 12, 32
    a, b = make_stub("hash", S), make_stub("hash2", S)
    first = [g % 2 for g in range(G)]
    na, nb_ = (_PackedProbe(a, S), _PackedProbe(b, S)) if packed else (a, b)
    eng = SelfPlayEngine(na, net2=nb_, size=S, n_games=G, sims=sims, energy=E, stop_exploration=0, num_moves=nm, symmetry="identity")
    assert eng.packed == packed
    eng.start_eval_games(np.arange(G), first_model=first)
    games = {gd["slot"]: gd for gd in eng.run()}
    assert len(games) == G and sum(eng.n_model_positions) == eng.status.total_evals
    assert abs(eng.n_model_positions[0] - eng.n_model_positions[1]) <= G * (sims + 1)
    eng.close()
    ref = {}
    for f in (0, 1):
        e1 = SelfPlayEngine(a, net2=b, size=S, n_games=1, sims=sims, energy=E, stop_exploration=0, num_moves=nm, symmetry="identity")
        e1.start_eval_games([0], first_model=[f])
        ref[f] = e1.run()[0]
        e1.close()
    for g in range(G):
        r = ref[first[g]]
        assert [m["action"] for m in games[g]["moves"]] == [m["action"] for m in r["moves"]], g
        assert games[g]["result"] == r["result"] and games[g]["modelB_name"] == r["modelB_name"]
        assert games[g]["modelB_name"] == ("hash_stub" if first[g] == 0 else "hash_stub_2")


class _SymNet(object):
    """Oracle-side wrapper: evaluates the stub through symmetry k exactly like random_symmetry_predict
    (symmetry.py:127-132), or the 8-fold average (build extension)."""

    def __init__(self, net, S, ks):
        self.net, self.S, self.ks, self.name = net, S, ks, net.name

    def predict_on_batch(self, boards):
        from oracle import oracle as ora
        pol, val = None, None
        for k in self.ks:
            p, v = self.net.predict_on_batch(ora.sym_board(k, boards))
            p = ora.sym_policy_inverse(self.S, k, p)
            pol = p if pol is None else pol + p
            val = v if val is None else val + v
        if len(self.ks) > 1:
            pol = pol / np.float32(len(self.ks))
            val = val / np.float32(len(self.ks))
        return pol, val


class _PackedProbe(object):
    """A stub net behind the resident net's PACKED-RECORD input contract (engine route: net.predict_packed -> sgo_stem_packed_dev),
    built so that the evaluation stays rounding-free: the real stem kernel runs with 0/1 weights that copy plane c of tap t into
    output channel 16 t + c (and the colour term into channel 144), the full S x S x 17 board tensor is re-assembled from those
    (S-2)^2 x 145 values -- every board point is under some tap of some output pixel -- and handed to the hash net.  Any wrong bit,
    tap, symmetry, colour flip or list index in the kernel changes the game."""
    packed_ok = True

    def __init__(self, net, S):
        import torch
        from sejonggo_amd import _lib
        self.net, self.name, self.S, self.t = net, net.name, S, S - 2
        self._lib, self.lib = _lib, _lib.require_gpu()
        w10 = torch.zeros(256, 10, 16)
        for t in range(9):
            for c in range(16):
                w10[16 * t + c, t, c] = 1.0
        self.w10 = w10.half().cuda().contiguous()
        self.bias = torch.zeros(256, dtype=torch.float16, device="cuda")
        wcol = torch.zeros(256)
        wcol[144] = 1.0                                   # relu(+1) = 1 black to play, relu(-1) = 0 white to play
        self.wcol = wcol.cuda().contiguous()
        ys = torch.arange(S)
        o = (ys - 1).clamp(0, S - 3)                      # an output pixel row whose window covers board row y, and the tap row
        d = ys - o
        self.oy, self.ox = o.cuda()[:, None].expand(S, S), o.cuda()[None, :].expand(S, S)
        self.tap = (d[:, None] * 3 + d[None, :]).cuda()

    def predict_packed(self, records_ptr, index_ptr, n, k=0, k_dev_ptr=None):
        import torch
        S, t = self.S, self.t
        y = torch.full((n, t, t, 256), 7.0, dtype=torch.float16, device="cuda")
        self._lib.check(self.lib.sgo_stem_packed_dev(S, n, records_ptr, index_ptr, int(k), k_dev_ptr, self.w10.data_ptr(),
                                                     self.bias.data_ptr(), self.wcol.data_ptr(), y.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "sgo_stem_packed_dev")
        capturing = torch.cuda.is_current_stream_capturing()      # captured rounds: no host read-backs inside the graph
        if not capturing:
            assert float(y[..., 145:].abs().max()) == 0.0
        g = y[:, self.oy, self.ox, :]                                                  # [n, S, S, 256]
        ch = (self.tap[..., None] * 16 + torch.arange(16, device="cuda"))[None].expand(n, S, S, 16)
        planes = torch.gather(g, 3, ch)
        col = (2.0 * y[:, 0, 0, 144] - 1.0)[:, None, None, None].expand(n, S, S, 1)
        X = torch.cat([planes, col.to(planes.dtype)], dim=3)
        if not capturing:
            assert bool(((X[..., :16] == 0) | (X[..., :16] == 1)).all())
        return self.net.predict_on_batch(X)

    def predict_on_batch(self, X):
        raise AssertionError("the engine must take the packed-record route for this net")


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "captured_rounds"])
@pytest.mark.parametrize("fn", ["async_02.npz", "async_05.npz", "async_07.npz", "async_08.npz"])
def test_packed_record_route_reproduces_the_golden_games(L, fn, graph):
    """The engine's packed-record route (evaluation list of record indices -> sgo_stem_packed_dev, no input tensor) against the
    reference's goldens: 9x9 and 5x5 games incl. the 'No best leaf' path, 19x19 at 400 sims and at 1 600 sims / 32-leaf rounds."""
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S = int(z["size"])
    net = _PackedProbe(make_stub(bytes(z["net"]).decode(), S), S)
    eng = _engine(z, net, graph=graph)
    assert eng.packed and eng.nn_in is None and eng.graph == graph
    games = eng.run()
    assert (eng.n_graph_replays > 0) == graph
    gd = games[0]
    assert len(gd["moves"]) == len(z["move_index"])
    for i, mv in enumerate(gd["moves"]):
        a = mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S
        assert a == z["move_index"][i] and mv["policy"].tobytes() == z["move_policy"][i].tobytes(), i
        assert mv["value"].tobytes() == z["move_value"][i].tobytes(), i
    assert gd["result"] == bytes(z["result"]).decode()
    assert eng.status.total_evals == int(z["n_predict"]) and eng.status.none_events == int(z["none_events"])
    eng.close()
    k = len(z["move_index"]) - 1
    eng = _engine(z, net, halt_at=k, graph=graph)
    eng.run()
    buf, nn, ne = eng.tree_serialize(0)
    assert nn == z["pm_n_nodes"][k] and hashlib.sha1(buf.tobytes()).digest()[:16] == z["pm_tree_hash"][k].tobytes()
    eng.close()


def test_two_half_populations_on_two_streams_equal_the_oracle(L):
    """engine.DualEngine: 48 games as two half-populations alternating on two HIP streams, every round a captured launch
    chain (stem from the records -> ... -> k_search -> k_compact -> board_advance), different round counts per game (some
    resign, some pass out), slots restarted mid-run.  Each game must equal the oracle's game for the same draws, move for
    move and tree for tree -- games do not interact, whatever the interleaving of the halves."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import DualEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 9, 48, 8, 48, 12
    net = make_stub("hash", S)
    rng = np.random.RandomState(31)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=2 * G)
    uni = rng.random_sample((2 * G, nm))
    resign = [None if rng.rand() < 0.6 else float(rng.uniform(-1, 0.2)) for _ in range(2 * G)]
    eng = DualEngine(_PackedProbe(net, S), n_games=G, size=S, sims=sims, energy=E, stop_exploration=5, num_moves=nm, komi=5.5,
                     symmetry="identity")
    assert eng.graph and eng.packed and len(eng.halves) == 2 and eng.halves[0].stream is not eng.halves[1].stream
    eng.start_games(np.arange(G), noises=noises[:G], uniforms=uni[:G], resign=resign[:G], ids=list(range(G)))
    done = {}
    slot_draw = {s: s for s in range(G)}
    nxt = G
    for _ in range(4000):
        st = eng.step()
        if st.n_done > 0:
            eng.drain()
            res = eng.results()
            restart = []
            for s in range(G):
                if res[s]["done"] == 1 and s in slot_draw:
                    done[slot_draw.pop(s)] = (eng.game_data(s, res[s]), res[s].copy(), eng.tree_serialize(s)[0].tobytes())
                    eng.records[s] = []
                    if nxt < 2 * G and s % 3 == 0:          # a third of the slots plays a second game
                        restart.append(s)
            if restart:
                draws = list(range(nxt, nxt + len(restart)))
                draws = [d for d in draws if d < 2 * G]
                restart = restart[:len(draws)]
                eng.start_games(restart, noises=noises[draws], uniforms=uni[draws], resign=[resign[d] for d in draws], ids=draws)
                for s, d in zip(restart, draws):
                    slot_draw[s] = d
                nxt += len(draws)
        if not slot_draw:
            break
    assert not slot_draw and len(done) >= G + 8
    assert all(e.n_graph_replays > 20 for e in eng.halves)
    for d, (gd, r, tree) in done.items():
        g = ora.Game(S, sims, E, 5, nm, uniforms=uni[d], noises=noises[d:d + 1], resign=resign[d]).run(net)
        o = g.result()
        assert g.n_moves == len(gd["moves"]) == r["n_moves"] and o["end_reason"] == r["end_reason"] and o["winner"] == r["winner"], d
        for i, mv in enumerate(gd["moves"]):
            m = g.move(i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes(), (d, i)
            assert mv["player"] == m["player"] and mv["value"].tobytes() == m["value"].tobytes(), (d, i)
        assert tree == g.tree_serialize()[0].tobytes(), d
    eng.close()


def test_two_stream_engine_refuses_large_rounds(L):
    """engine.DualEngine is for short rounds: above MAX_ROUND_PIXELS leaf pixels per round it is refused at construction (long
    kernels on two streams have stalled captured rounds: DESIGN.md section 5), before any context is allocated."""
    from sejonggo_amd.engine import DualEngine
    from sejonggo_amd.stub_nets import make_stub
    with pytest.raises(ValueError):
        DualEngine(make_stub("hash", 19), n_games=1024, size=19, sims=400, energy=8)
    with pytest.raises(ValueError):
        DualEngine(make_stub("hash", 19), n_games=1)
    assert 256 * 8 * 49 < DualEngine.MAX_ROUND_PIXELS < 512 * 8 * 289          # config 2 inside, 19x19 / 512 games outside


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5, 6, 7, "avg8"])
def test_symmetry_modes_equal_the_oracle(L, mode, packed):
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, nm = 9, 32, 8, 6
    net = make_stub("hash", S)
    rng = np.random.RandomState(21)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=2)
    uni = rng.random_sample((2, nm))
    eng = SelfPlayEngine(_PackedProbe(net, S) if packed else net, size=S, n_games=2, sims=sims, energy=E, stop_exploration=3,
                         num_moves=nm, komi=5.5, symmetry=mode)
    assert eng.packed == packed                               # packed: the stem kernel applies the symmetry (eight passes for avg8)
    eng.start_games([0, 1], noises=noises, uniforms=uni)
    games = eng.run()
    ks = list(range(8)) if mode == "avg8" else [mode]
    for gd in games:
        s = gd["slot"]
        g = ora.Game(S, sims, E, 3, nm, uniforms=uni[s], noises=noises[s:s + 1]).run(_SymNet(net, S, ks))
        assert [m["move"][0] + S * m["move"][1] if m["move"][1] != S else S * S for m in gd["moves"]] == \
               [g.move(i)["action"] for i in range(g.n_moves)]
        ta, _, _ = eng.tree_serialize(s)
        tb, _, _ = g.tree_serialize()
        assert ta.tobytes() == tb.tobytes()
    eng.close()


def test_real_net_selfplay_runs_and_restarts(L):
    """Random-init fp16 resnet: games finish, slots restart, records are well-formed."""
    import torch
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.net import build_net
    S = 9
    net = build_net(S, 2, 64, name="tiny")
    eng = SelfPlayEngine(net, size=S, n_games=32, sims=16, energy=8, stop_exploration=4, num_moves=12, symmetry="random1")
    eng.start_games(np.arange(32))
    games = eng.run()
    assert len(games) == 32
    for gd in games:
        assert 1 <= len(gd["moves"]) <= 12
        for mv in gd["moves"]:
            assert mv["board"].shape == (1, S, S, 17) and abs(float(mv["value"])) <= 1.0
            assert mv["policy"].shape == (S * S + 1,) and mv["policy"].min() >= 0
    eng.start_games(np.arange(8))      # restart some slots
    games2 = eng.run()
    assert len(games2) == 32
    eng.close()


@pytest.mark.parametrize("fn", ["async_02.npz", "async_05.npz", "async_07.npz", "async_13.npz"])
def test_golden_games_from_the_shared_block_pool(L, fn):
    """The context-wide block pool: with a private region of the minimum size (energy + 2 blocks) practically every tree block
    of the game is an overflow id backed by the shared pool -- taken one at a time inside k_search, handed back by the re-root's
    mark / rebuild, merged into the free stack by k_compact.  The reference's goldens must come out move for move, with the
    whole-tree hash after the last search (block ids never enter a serialised tree)."""
    from sejonggo_amd.stub_nets import make_stub
    z = load(fn)
    S, E, sims = int(z["size"]), int(z["energy"]), int(z["sims"])
    net = make_stub(bytes(z["net"]).decode(), S)
    kw = dict(blocks_per_game=E + 2, shared_blocks=6 * sims + 64)
    eng = _engine(z, net, **kw)
    info = eng.pool_info()
    assert info["private_per_game"] == E + 2 and info["shared_blocks"] == 6 * sims + 64 and info["ids_per_game"] > E + 2
    games = eng.run()
    gd = games[0]
    assert len(gd["moves"]) == len(z["move_index"])
    for i, mv in enumerate(gd["moves"]):
        a = mv["move"][0] + S * mv["move"][1] if mv["move"][1] != S else S * S
        assert a == z["move_index"][i] and mv["policy"].tobytes() == z["move_policy"][i].tobytes(), i
        assert mv["value"].tobytes() == z["move_value"][i].tobytes(), i
    assert gd["result"] == bytes(z["result"]).decode()
    assert eng.status.total_evals == int(z["n_predict"]) and eng.status.none_events == int(z["none_events"])
    after = eng.pool_info()
    assert after["shared_free_low_water"] < after["shared_blocks"] - sims // 2            # the pool was really used ...
    assert gd["blocks_high_water"] > E + 2
    eng.start_games([0], noises=z["noises"][:1], uniforms=np.zeros((1, max(1, eng.max_moves))))
    eng.step()
    assert eng.pool_info()["shared_free"] >= after["shared_blocks"] - 2               # ... and a restart hands all of it back
    eng.close()
    k = len(z["move_index"]) - 1
    eng = _engine(z, net, halt_at=k, **kw)
    eng.run()
    buf, nn, ne = eng.tree_serialize(0)
    assert nn == z["pm_n_nodes"][k] and hashlib.sha1(buf.tobytes()).digest()[:16] == z["pm_tree_hash"][k].tobytes()
    t = eng.root_table(0)
    assert int(t["N"].sum()) > 0 and eng.board(0).shape == (1, S, S, 17)
    eng.close()


def test_many_games_share_one_pool_and_exhaustion_is_loud(L):
    """32 concurrent games with 10-block private regions on one shared pool: (a) a pool that is large enough -- every game
    equals the oracle's, whole trees included, although nearly all of their blocks are shared ones taken and returned move by
    move; (b) a pool that is too small -- games fail with SGO_ERR_CAPACITY, loudly, whoever finishes still equals the oracle's,
    and nothing hangs."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, G, nm = 9, 64, 8, 32, 10
    net = make_stub("hash", S)
    rng = np.random.RandomState(41)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    for pool, expect_fail in ((G * 220, False), (G * 40, True)):
        eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=4, num_moves=nm, komi=5.5,
                             symmetry="identity", blocks_per_game=10, shared_blocks=pool, raise_on_error=False)
        eng.start_games(np.arange(G), noises=noises, uniforms=uni)
        for _ in range(2000):
            st = eng.step()
            if st.n_records >= G:
                eng.drain()
            if st.n_active == 0:
                break
        assert st.n_active == 0
        eng.drain()
        res = eng.results()
        failed = [s for s in range(G) if res[s]["done"] < 0]
        assert (len(failed) > 0) == expect_fail and all(res[s]["done"] == -201 for s in failed)
        ok = [s for s in range(G) if res[s]["done"] == 1]
        assert len(ok) + len(failed) == G and (expect_fail or len(ok) == G)
        for s in ok:
            g = ora.Game(S, sims, E, 4, nm, uniforms=uni[s], noises=noises[s:s + 1]).run(net)
            moves = eng.records[s]
            assert g.n_moves == len(moves) == nm, s
            for i, mv in enumerate(moves):
                m = g.move(i)
                assert mv["action"] == m["action"] and mv["policy"].tobytes() == m["policy"].tobytes(), (s, i)
            ta, _, _ = eng.tree_serialize(s)
            tb, _, _ = g.tree_serialize()
            assert ta.tobytes() == tb.tobytes(), s
        info = eng.pool_info()
        assert 0 <= info["shared_free_low_water"] < pool
        if not expect_fail:
            assert max(int(r["blocks_high_water"]) for r in res) > 60          # trees several times the private region
        eng.close()


def test_capacity_error_is_loud(L):
    from sejonggo_amd import _lib
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    eng = SelfPlayEngine(make_stub("hash", 9), size=9, n_games=1, sims=64, energy=8, stop_exploration=0, num_moves=4,
                         blocks_per_game=20, symmetry="identity")
    eng.start_games([0])
    with pytest.raises(_lib.SgoError):
        eng.run()
    eng.close()


def test_fused_inference_net_matches_fp32_reference(L):
    """The hot-loop net (padded input, hand-written bias/skip/ReLU epilogue, merged head GEMM) against the
    plain PyTorch fp32 forward of the same weights.  Tolerance for the fp16 path:
    max-abs dp <= 2e-3, dv <= 5e-3."""
    import torch
    from sejonggo_amd.net import build_fused_net
    for S, nb, ch in ((9, 3, 64), (19, 2, 256)):
        fnet, ref = build_fused_net(S, nb, ch, seed=3)
        ref = ref.cuda().float()
        X = torch.zeros((16, S, S, 17), device="cuda")
        X[..., :16] = (torch.rand((16, S, S, 16), device="cuda") < 0.2).float()
        X[..., 16] = 1.0
        X[8:, :, :, 16] = -1.0
        p0, v0 = ref.predict_on_batch(X)
        p1, v1 = fnet.predict_on_batch(X.half())
        assert p1.shape == (16, S * S + 1) and v1.shape == (16, 1)
        assert float((p0 - p1).abs().max()) <= 2e-3, float((p0 - p1).abs().max())
        assert float((v0 - v1).abs().max()) <= 5e-3, float((v0 - v1).abs().max())
        Xp = torch.nn.functional.pad(X.half(), (0, 15))
        p2, v2 = fnet.predict_on_batch(Xp)
        # same input, padded by the caller: equal up to the library's choice of conv algorithm per call
        assert float((p1 - p2).abs().max()) <= 1e-3 and float((v1 - v2).abs().max()) <= 2e-3


def test_nn_pack_channel_padded_layout(L):
    import torch
    lib = L.load()
    S, n = 19, 5
    RW = lib.sgo_packed_words(S)
    boards = torch.zeros((n, S, S, 17), dtype=torch.int32, device="cuda")
    boards[..., :16] = (torch.rand((n, S, S, 16), device="cuda") < 0.3).int()
    boards[..., 16] = -1
    packed = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    st = L.stream_ptr()
    L.check(lib.sgo_pack_dev(S, n, L.ptr(boards), L.ptr(packed), st))
    out = torch.full((n, S, S, 32), 7.0, dtype=torch.float16, device="cuda")
    L.check(lib.sgo_nn_pack_dev(S, n, L.ptr(packed), None, 0, 2, 0, L.ptr(out), st))
    assert torch.equal(out[..., :17].int(), boards) and float(out[..., 17:].abs().max()) == 0.0


def test_engine_with_fused_net(L):
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.net import build_fused_net
    fnet, _ = build_fused_net(9, 2, 64)
    eng = SelfPlayEngine(fnet, size=9, n_games=16, sims=16, energy=8, stop_exploration=2, num_moves=4)
    assert eng.layout == 2
    eng.start_games(np.arange(16))
    games = eng.run()
    assert len(games) == 16 and all(len(g["moves"]) == 4 for g in games)
    eng.close()


def test_resign_and_no_noise_paths_equal_the_oracle(L):
    """resign thresholds (nomodel_self_play.py:170-173) and self_play=False (no Dirichlet mix, float32 root)."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    S, sims, E, nm, G = 9, 32, 8, 30, 12
    net = make_stub("hash", S)
    rng = np.random.RandomState(5)
    uni = rng.random_sample((G, nm))
    resign = [None, -0.2, 0.0, 0.3, -0.5, 0.9, None, -0.1, 0.1, 0.5, -0.9, 0.7]
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=4, num_moves=nm, komi=5.5,
                         symmetry="identity", self_play=False)
    eng.start_games(np.arange(G), uniforms=uni, resign=resign)
    games = {gd["slot"]: gd for gd in eng.run()}
    res = eng.results()
    n_resigned = 0
    for s in range(G):
        g = ora.Game(S, sims, E, 4, nm, self_play=False, uniforms=uni[s], resign=resign[s]).run(net)
        r = g.result()
        assert r["end_reason"] == res[s]["end_reason"], s
        n_resigned += r["end_reason"] == 1
        assert g.n_moves == len(games[s]["moves"]) == res[s]["n_moves"], s
        assert r["winner"] == res[s]["winner"] and r["black"] == res[s]["black"] and r["white"] == res[s]["white"]
        assert r["last_player"] == res[s]["last_player"]
        for i, mv in enumerate(games[s]["moves"]):
            m = g.move(i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes()
        if r["end_reason"] == 1:
            assert games[s]["result"] == "%s+R" % {1: "B", -1: "W"}[r["last_player"]]
    assert n_resigned >= 3
    eng.close()


@pytest.mark.parametrize("halves", [1, 2], ids=["one_population", "two_half_populations"])
def test_selfplay_worker_body_writes_samples(L, tmp_path, halves):
    """run_selfplay: directory reservation, engine loop with slot restarts, sample files in the reference's layout -- on the
    single-context engine (a small net on the tensor route) and on engine.DualEngine (conf['ENGINE_HALVES'] = 2: the resident
    net's packed-record route, captured rounds, two streams; slots of both halves finish, restart and write)."""
    import os
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd import sgfsave
    from sejonggo_amd.conf import conf
    from sejonggo_amd.net import build_fused_net
    from sejonggo_amd.selfplay_worker import run_selfplay
    keep = dict(conf)
    try:
        conf.update({'SIZE': 9, 'MCTS_SIMULATIONS': 16, 'ENERGY': 8, 'STOP_EXPLORATION': 2, 'N_GAMES': 6,
                     'SELF_PLAY_DIR': str(tmp_path / "sp"), 'GAMES_PER_GPU': 4, 'ENGINE_HALVES': halves})
        fnet, _ = build_fused_net(9, 1, 32 if halves == 1 else 256, name="wk")
        pq.set_model_factory(lambda kind: fnet)
        seen = []
        played = run_selfplay(0, "BEST_SYM", n_games=6, games_per_gpu=4, on_game=lambda g, gd: seen.append((g, len(gd['moves']))),
                              max_steps=3000)
        assert played == len(seen) and played >= 4
        for g, n in seen:
            d = os.path.join(conf['SELF_PLAY_DIR'], "wk", "game_%05d" % g)
            assert len(os.listdir(d)) == n
            assert os.path.isfile(os.path.join(d, "move_000", "sample.h5"))
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
        conf.clear()
        conf.update(keep)


def test_writer_processes_write_the_same_files_as_writer_threads(L, tmp_path):
    """conf['WRITER_PROCESSES']: finished games go to spawned torch-free writer processes as packed records; same seeds, same
    games, so every sample file must hold the same three arrays as the writer-thread run's."""
    import os
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    from sejonggo_amd.selfplay_worker import run_selfplay
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import read_sample
    keep = dict(conf)
    net = make_stub("hash", 9)
    out = {}
    try:
        for mode, procs in (("threads", 0), ("procs", 2)):
            conf.update({'SIZE': 9, 'MCTS_SIMULATIONS': 16, 'ENERGY': 8, 'STOP_EXPLORATION': 0, 'N_GAMES': 6, 'NUM_MOVES': 7,
                         'SELF_PLAY_DIR': str(tmp_path / mode), 'GAMES_PER_GPU': 3, 'WRITER_PROCESSES': procs,
                         'RESIGNATION_PERCENT': 1.0})
            pq.set_model_factory(lambda kind: net)
            stats = {}
            played = run_selfplay(0, "BEST", n_games=6, games_per_gpu=3, max_steps=400, stats=stats,
                                  engine_kwargs={'num_moves': 7, 'dirichlet_epsilon': 0.0})
            assert played == 6 and stats["files"] == stats["moves"] == 42
            files = {}
            root = os.path.join(conf['SELF_PLAY_DIR'], net.name)
            for g in sorted(os.listdir(root)):
                for m in sorted(os.listdir(os.path.join(root, g))):
                    files[(g, m)] = read_sample(os.path.join(root, g, m, "sample.h5"))
            out[mode] = files
            pq.destroy_predicting_workers([0])
        assert out["threads"].keys() == out["procs"].keys() and len(out["procs"]) == 42
        for k in out["threads"]:
            for a, b in zip(out["threads"][k], out["procs"][k]):
                assert np.asarray(a).tobytes() == np.asarray(b).tobytes(), k
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
        conf.clear()
        conf.update(keep)


def test_tree_dict_view_equals_the_canonical_serialisation(L):
    """engine.tree_dict rebuilds the reference's nested dict nodes from the device tree; hashing that dict tree the
    way the golden harness hashes the reference's dict tree must reproduce the golden hash."""
    import hashlib
    from sejonggo_amd.mcts1 import TreeNode
    from sejonggo_amd.stub_nets import make_stub
    from tests.helpers import dict_tree_hash
    z = load("async_02.npz")
    S = int(z["size"])
    eng = _engine(z, make_stub("hash", S), halt_at=3)
    eng.run()
    tree = eng.tree_dict(0)
    h, nn = dict_tree_hash(tree)
    assert nn == z["pm_n_nodes"][3] and h == z["pm_tree_hash"][3].tobytes()
    assert tree['count'] == z["pm_root_count"][3] and set(tree['subtree']) == set(np.flatnonzero(z["pm_EX"][3]))
    some = next(iter(tree['subtree'].values()))
    assert set(some) == {'index', 'count', 'value', 'mean_value', 'p', 'subtree', 'parent', 'virtual_loss'}
    assert some['parent'] is tree
    view = TreeNode(tree)
    assert view.v == tree['count'] and len(view.children) == len(tree['subtree'])
    assert view.best_move().move == int(np.argmax(z["pm_N"][3]))
    eng.close()


@pytest.mark.parametrize("S,sims,E,nm", [(7, 36, 4, 10), (13, 50, 8, 6), (9, 100, 32, 5), (9, 130, 64, 4), (5, 30, 16, 20)])
def test_other_sizes_and_energies_equal_the_oracle(L, S, sims, E, nm):
    """Board sizes without reference goldens (7, 13), wide leaf batches (32, 64) and sims not divisible by energy:
    the (pinned) oracle is the checker."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    net = make_stub("hash", S)
    rng = np.random.RandomState(S * 1000 + E)
    G = 6
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=3, num_moves=nm, komi=6.5,
                         symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = {gd["slot"]: gd for gd in eng.run()}
    want_evals = want_none = 0
    for s in range(G):
        g = ora.Game(S, sims, E, 3, nm, komi=6.5, uniforms=uni[s], noises=noises[s:s + 1]).run(net)
        c = g.counters()
        want_evals += c["n_predict"]
        want_none += c["none_events"]
        assert g.n_moves == len(games[s]["moves"]), s
        for i, mv in enumerate(games[s]["moves"]):
            m = g.move(i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes(), (s, i)
        ta, _, _ = eng.tree_serialize(s)
        tb, _, _ = g.tree_serialize()
        assert ta.tobytes() == tb.tobytes(), s
        r = g.result()
        assert games[s]["black_points"] == r["black"] and games[s]["white_points"] == r["white"]
    # network evaluations consumed and "No best leaf" events: the oracle's own counters, summed over the games
    assert eng.status.total_evals == want_evals and eng.status.none_events == want_none
    eng.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108])
def test_fuzzed_configurations_equal_the_oracle(L, seed):
    """Randomly drawn configurations (board size, simulations not divisible by the energy, energies 1..32, exploration
    cut-off, komi, self-play with Dirichlet noise or evaluation mode, per-game resign thresholds, stub net): every move,
    board, prior vector, value, result and the final tree must equal the oracle's, byte for byte."""
    from oracle import oracle as ora
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    rng = np.random.RandomState(seed)
    S = int(rng.choice([5, 5, 7, 9, 9, 13]))
    E = int(rng.choice([1, 2, 4, 8, 16, 32]))
    sims = E * int(rng.randint(2, 7)) + int(rng.randint(0, E))
    nm = int(rng.randint(4, {5: 40, 7: 24, 9: 16, 13: 8}[S]))
    stop = int(rng.randint(0, nm + 1))
    komi = float(rng.choice([0.5, 5.5, 7.5]))
    self_play = bool(rng.randint(0, 2))
    net = make_stub(str(rng.choice(["hash", "uniform"])), S)
    G = 8
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    resign = [None if rng.rand() < 0.5 else float(rng.uniform(-1, 1)) for _ in range(G)]
    # odd seeds: board tensors through k_nn_pack, eager steps; even seeds: packed records through the stem kernel, captured rounds
    route = _PackedProbe(net, S) if seed % 2 == 0 else net
    eng = SelfPlayEngine(route, size=S, n_games=G, sims=sims, energy=E, stop_exploration=stop, num_moves=nm, komi=komi,
                         symmetry="identity", self_play=self_play, graph=(seed % 2 == 0))
    assert eng.graph == (seed % 2 == 0)
    eng.start_games(np.arange(G), noises=noises if self_play else None, uniforms=uni, resign=resign)
    games = {gd["slot"]: gd for gd in eng.run()}
    res = eng.results()
    cfg = (S, E, sims, nm, stop, komi, self_play)
    for s in range(G):
        g = ora.Game(S, sims, E, stop, nm, self_play=self_play, komi=komi, uniforms=uni[s],
                     noises=noises[s:s + 1] if self_play else None, resign=resign[s]).run(net)
        r = g.result()
        assert r["end_reason"] == res[s]["end_reason"] and g.n_moves == res[s]["n_moves"], (cfg, s)
        assert r["winner"] == res[s]["winner"] and r["black"] == res[s]["black"] and r["white"] == res[s]["white"], (cfg, s)
        moves = games[s]["moves"] if s in games else []
        assert len(moves) == g.n_moves, (cfg, s)
        for i, mv in enumerate(moves):
            m = g.move(i)
            assert np.array_equal(mv["board"], m["board"]) and mv["policy"].tobytes() == m["policy"].tobytes(), (cfg, s, i)
            assert mv["player"] == m["player"] and mv["value"].tobytes() == m["value"].tobytes(), (cfg, s, i)
        ta, na, _ = eng.tree_serialize(s)
        tb, nb, _ = g.tree_serialize()
        assert na == nb and ta.tobytes() == tb.tobytes(), (cfg, s)
    eng.close()


def test_worker_survives_a_slot_that_outgrows_its_block_pool(L, tmp_path, monkeypatch):
    """With a deliberately tiny block pool some games fail with SGO_ERR_CAPACITY: the worker body must discard exactly
    those games (loudly), keep the others, and terminate."""
    import os
    from sejonggo_amd import engine as eng_mod, predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    from sejonggo_amd.selfplay_worker import run_selfplay
    from sejonggo_amd.stub_nets import make_stub
    keep = dict(conf)
    real = eng_mod.SelfPlayEngine

    def tiny(*a, **k):
        k["blocks_per_game"] = 40
        k["num_moves"] = 6
        return real(*a, **k)

    monkeypatch.setattr(eng_mod, "SelfPlayEngine", tiny)
    try:
        conf.update({'SIZE': 9, 'MCTS_SIMULATIONS': 24, 'ENERGY': 8, 'STOP_EXPLORATION': 0, 'N_GAMES': 6,
                     'SELF_PLAY_DIR': str(tmp_path / "sp"), 'GAMES_PER_GPU': 3})
        net = make_stub("hash", 9)
        pq.set_model_factory(lambda kind: net)
        seen = []
        played = run_selfplay(0, "BEST", n_games=6, games_per_gpu=3, on_game=lambda g, gd: seen.append(g), max_steps=2000)
        root = os.path.join(conf['SELF_PLAY_DIR'], net.name)
        kept = sorted(os.listdir(root)) if os.path.isdir(root) else []
        assert played == len(seen) == len(kept)          # failed games leave no directory behind
        assert played < 6                                # the pool really was too small for some of them
    finally:
        pq.set_model_factory(None)
        pq.destroy_predicting_workers([0])
        conf.clear()
        conf.update(keep)


def test_fused_conv_kernel_matches_torch(L):
    """sgo_conv3x3_bias_act_dev (dispatch to the hand-written tower / stem kernels, bias / skip / ReLU fused) against torch
    conv2d in fp32; shapes without a hand-written kernel are refused loudly (SGO_ERR_UNSUPPORTED), never computed elsewhere."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    torch.manual_seed(0)
    x = torch.zeros(3, 7, 7, 64, device="cuda", dtype=torch.float16)
    rc = lib.sgo_conv3x3_bias_act_dev(3, 7, 7, 64, 64, 1, x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == L.SGO_ERR_UNSUPPORTED and b"no hand-written kernel" in lib.sgo_last_error()
    for (n, h, c, k, pad, with_skip) in [(8, 17, 256, 256, 1, True), (8, 17, 256, 256, 1, False), (5, 19, 32, 256, 0, False),
                                         (1, 9, 32, 256, 0, False), (300, 19, 32, 256, 0, False), (37, 5, 32, 256, 0, False),
                                         (300, 17, 256, 256, 1, True),
                                         (15000, 17, 256, 256, 1, True)]:     # > 2^31 bytes per tensor: sliced launches
        x = (torch.randn(n, c, h, h, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(k, c, 3, 3, device="cuda") * 0.03).half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(k, device="cuda").half()
        ho = h + 2 * pad - 2
        skip = (torch.randn(n, k, ho, ho, device="cuda")).half().contiguous(memory_format=torch.channels_last) if with_skip else None
        y = torch.empty((n, k, ho, ho), dtype=torch.float16, device="cuda", memory_format=torch.channels_last)
        L.check(lib.sgo_conv3x3_bias_act_dev(n, h, h, c, k, pad, x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                             None if skip is None else skip.data_ptr(), y.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
        err, top = 0.0, 1.0
        for o in range(0, n, 2048):        # reference in slices to bound fp32 memory
            ref = F.conv2d(x[o:o + 2048].float(), w.float(), b.float(), padding=pad)
            if with_skip:
                ref = ref + skip[o:o + 2048].float()
            ref = torch.relu(ref)
            err = max(err, float((y[o:o + 2048].float() - ref).abs().max()))
            top = max(top, float(ref.abs().max()))
        assert err <= 2e-2 * top, (n, h, c, k, pad, with_skip, err)


def test_stem_conv_kernel(L):
    """sgo_conv3x3_stem_dev, the hand-written stem kernel (csrc/sgo_stem.hpp; 'valid' 3x3, 17 planes in 32 channels -> 256):
    against torch in fp32 within fp16 output rounding, exact on small-integer data, bit-identical across relaunches and
    slice sizes, on the real input format (bit planes + the colour plane, channels 17..31 zero)."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(3)
    for (n, h, wd) in [(1, 19, 19), (2, 9, 9), (7, 5, 5), (64, 19, 19), (333, 19, 19), (100, 13, 7)]:
        x = torch.zeros(n, h, wd, 32, device="cuda", dtype=torch.float16)
        x[..., :16] = (torch.rand(n, h, wd, 16, device="cuda") < 0.3).half()
        x[..., 16] = torch.where(torch.rand(n, 1, 1, device="cuda") < 0.5, 1.0, -1.0).half()
        w = (torch.randn(256, 3, 3, 32, device="cuda") * 0.1).half()
        b = torch.randn(256, device="cuda").half()
        y = torch.full((n, h - 2, wd - 2, 256), 9.0, device="cuda", dtype=torch.float16)
        L.check(lib.sgo_conv3x3_stem_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st))
        ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float()).permute(0, 2, 3, 1))
        err = (y.float() - ref).abs()
        assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (n, h, wd, float(err.max()))
        for cap in (0, 1, 5):
            lib.sgo_conv_tower_slice_cap(cap)
            try:
                y2 = torch.full_like(y, 4.0)
                L.check(lib.sgo_conv3x3_bias_act_dev(n, h, wd, 32, 256, 0, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y2.data_ptr(), st))
            finally:
                lib.sgo_conv_tower_slice_cap(0)
            assert torch.equal(y, y2), (n, h, wd, cap)
    # exact: integer weights / bias, 0/1 inputs -> every partial sum is an integer well inside fp16's exact range
    n, h, wd = 50, 19, 19
    x = torch.zeros(n, h, wd, 32, device="cuda", dtype=torch.float16)
    x[..., :17] = torch.randint(0, 2, (n, h, wd, 17), device="cuda").half()
    w = torch.randint(-3, 4, (256, 3, 3, 32), device="cuda").half()
    b = torch.randint(-5, 6, (256,), device="cuda").half()
    y = torch.empty((n, h - 2, wd - 2, 256), device="cuda", dtype=torch.float16)
    L.check(lib.sgo_conv3x3_stem_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st))
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float()).permute(0, 2, 3, 1))
    assert torch.equal(y.float(), ref)


@pytest.mark.parametrize("S", [5, 7, 9, 13, 19])
def test_stem_packed_kernel(L, S):
    """sgo_stem_packed_dev (csrc/sgo_stem_packed.hpp): the stem read straight from packed position records.  Reference = torch
    conv2d in fp32 on the 17-plane tensor that k_nn_pack makes of the same records under the same symmetry (that kernel is
    pinned against the reference's transforms, test_nn_pack_all_symmetries).  Positions come from seeded random playouts (both
    sides to move, dense and sparse boards), rows are addressed through a shuffled index list and densely; all 8 symmetries;
    EXACT on integer weights (any wrong bit, tap, plane flip or symmetry shows), within fp16 rounding on random weights,
    identical with the symmetry passed by value and through device memory, ragged batch sizes around the 256-pixel tiles."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    st = L.stream_ptr()
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + S)
    n = 700 if S <= 9 else 150
    cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    nxt = torch.zeros_like(cur)
    legal = torch.full((n, NW), -1, dtype=torch.int32, device="cuda")
    legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1
    shifts = torch.arange(32, device="cuda", dtype=torch.int32)
    plies = torch.randint(0, 2 * S * S // 3, (n,), device="cuda", generator=g)
    for ply in range(int(plies.max().item()) + 1):               # record i stops after plies[i] moves (pass from then on)
        bits = ((legal.unsqueeze(-1) >> shifts) & 1).reshape(n, NW * 32)[:, :A].float()
        bits[:, A - 1] = 0.02
        mv = torch.multinomial(bits, 1, generator=g).reshape(-1).to(torch.int32)
        mv = torch.where(plies > ply, mv, torch.full_like(mv, A - 1))          # finished records pass (and are kept as they are)
        keep = cur.clone()
        L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
        live = (plies > ply)[:, None]
        cur = torch.where(live, nxt, keep)
    torch.manual_seed(5 + S)
    t = S - 2
    for trial, nn in enumerate([n, 1, 2, max(1, 256 // (t * t)), 256 // (t * t) + 1, 37]):
        nn = min(nn, n)
        idx = torch.randperm(n, device="cuda", generator=g)[:nn].to(torch.int32).contiguous()
        integer = trial % 2 == 0
        if integer:
            w = torch.randint(-3, 4, (256, 17, 3, 3), device="cuda").float()
            b = torch.randint(-5, 6, (256,), device="cuda").float()
        else:
            w = (torch.randn(256, 17, 3, 3, device="cuda") * 0.1).half().float()
            b = torch.randn(256, device="cuda").half().float()
        w10 = torch.zeros(256, 10, 16, device="cuda")
        w10[:, :9] = w[:, :16].permute(0, 2, 3, 1).reshape(256, 9, 16)
        w10 = w10.half().contiguous()
        wcol = w[:, 16].reshape(256, 9).sum(dim=1).contiguous()
        bh = b.half().contiguous()
        for k in range(8):
            x = torch.zeros((nn, S, S, 17), dtype=torch.float32, device="cuda")
            L.check(lib.sgo_nn_pack_dev(S, nn, L.ptr(cur), L.ptr(idx), k, 0, 1, L.ptr(x), st))
            ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2), w, b)).permute(0, 2, 3, 1)
            y = torch.full((nn, t, t, 256), 9.0, dtype=torch.float16, device="cuda")
            L.check(lib.sgo_stem_packed_dev(S, nn, L.ptr(cur), L.ptr(idx), k, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(y), st))
            if integer:
                assert float(ref.max()) < 2048 and torch.equal(y.float(), ref), (S, nn, k)
            else:
                err = (y.float() - ref).abs()
                assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (S, nn, k, float(err.max()))
            kd = torch.tensor([k], dtype=torch.int32, device="cuda")
            y2 = torch.full_like(y, 3.0)
            L.check(lib.sgo_stem_packed_dev(S, nn, L.ptr(cur), L.ptr(idx), 0, L.ptr(kd), L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(y2), st))
            assert torch.equal(y, y2), (S, nn, k)
        if nn == n:                                               # dense form (no index list) = the identity list
            ar = torch.arange(n, dtype=torch.int32, device="cuda")
            ya, yb = torch.empty((n, t, t, 256), dtype=torch.float16, device="cuda"), torch.empty((n, t, t, 256), dtype=torch.float16, device="cuda")
            L.check(lib.sgo_stem_packed_dev(S, n, L.ptr(cur), None, 3, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(ya), st))
            L.check(lib.sgo_stem_packed_dev(S, n, L.ptr(cur), L.ptr(ar), 3, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(yb), st))
            assert torch.equal(ya, yb)
    assert lib.sgo_stem_packed_dev(S, 1, L.ptr(cur), None, 8, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(y), st) < 0    # bad symmetry
    assert lib.sgo_stem_packed_dev(6, 1, L.ptr(cur), None, 0, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(y), st) < 0    # bad size


def test_stem_packed_kernel_at_the_headline_batch(L):
    """8 192 records x 19 x 19 (the bench's launch) and 32 768 (config 5's): sampled rows against the 17-plane fp32 reference."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    S, t = 19, 17
    st = L.stream_ptr()
    RW = lib.sgo_packed_words(S)
    torch.manual_seed(2)
    base = 4096
    boards = torch.zeros((base, S, S, 17), dtype=torch.int32, device="cuda")
    occ = torch.rand((base, S, S, 1), device="cuda")
    colour = torch.rand((base, S, S, 8), device="cuda") < 0.5
    dens = torch.rand((base, 1, 1, 1), device="cuda") * 0.7
    stone = occ < dens
    boards[..., 0:16:2] = (stone & colour).int()
    boards[..., 1:16:2] = (stone & ~colour).int()
    boards[..., 16] = torch.where(torch.rand((base, 1, 1), device="cuda") < 0.5, 1, -1).int()
    recs = torch.zeros((base, RW), dtype=torch.int32, device="cuda")
    L.check(lib.sgo_pack_dev(S, base, L.ptr(boards), L.ptr(recs), st))
    w = (torch.randn(256, 17, 3, 3, device="cuda") * 0.1).half().float()
    b = torch.randn(256, device="cuda").half().float()
    w10 = torch.zeros(256, 10, 16, device="cuda")
    w10[:, :9] = w[:, :16].permute(0, 2, 3, 1).reshape(256, 9, 16)
    w10, wcol, bh = w10.half().contiguous(), w[:, 16].reshape(256, 9).sum(dim=1).contiguous(), b.half().contiguous()
    for n in (8192, 32768):
        idx = torch.randint(0, base, (n,), device="cuda").to(torch.int32)
        y = torch.full((n, t, t, 256), 9.0, dtype=torch.float16, device="cuda")
        L.check(lib.sgo_stem_packed_dev(S, n, L.ptr(recs), L.ptr(idx), 5, None, L.ptr(w10), L.ptr(bh), L.ptr(wcol), L.ptr(y), st))
        rows = sorted(set([0, 1, n - 1, n - 2] + [int(v) for v in np.random.RandomState(n).randint(0, n, size=40)]))
        ri = torch.tensor(rows, device="cuda")
        x = torch.zeros((len(rows), S, S, 17), dtype=torch.float32, device="cuda")
        sub = idx[ri].contiguous()
        L.check(lib.sgo_nn_pack_dev(S, len(rows), L.ptr(recs), L.ptr(sub), 5, 0, 1, L.ptr(x), st))
        ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2), w, b)).permute(0, 2, 3, 1)
        err = (y[ri].float() - ref).abs()
        assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (n, float(err.max()))
        assert not bool((y == 9.0).all(dim=-1).any())            # every pixel of every row was written
        del y


@pytest.fixture(params=[0, 1], ids=["k_conv8w", "k_conv4w"])
def tower_kernel(request, L):
    """Both hand-written tower kernels (sgo_conv8w.hpp: one 512-thread workgroup per CU; sgo_conv4w.hpp: two 256-thread
    workgroups per CU) go through the same parity tests."""
    lib = L.load()
    old = lib.sgo_conv_tower_kernel(request.param)
    yield request.param
    lib.sgo_conv_tower_kernel(old)


def test_tower_kernels_agree_bit_for_bit(L):
    import torch
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(11)
    for (n, h, wd, with_skip) in [(3, 17, 17, True), (64, 7, 7, False), (7, 5, 19, True), (1500, 17, 17, True), (2, 19, 19, False)]:
        x = torch.relu(torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
        w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
        b = torch.randn(256, device="cuda").half()
        skip = torch.randn(n, h, wd, 256, device="cuda").half() if with_skip else None
        outs = []
        for kern in (0, 1):
            old = lib.sgo_conv_tower_kernel(kern)
            try:
                y = torch.full((n, h, wd, 256), 7.0, device="cuda", dtype=torch.float16)
                L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                  None if skip is None else skip.data_ptr(), y.data_ptr(), st))
                outs.append(y)
            finally:
                lib.sgo_conv_tower_kernel(old)
        assert torch.equal(outs[0], outs[1]), (n, h, wd, with_skip)       # same MFMA order per output: identical bits


def test_packed_tower_kernel(L):
    """sgo_conv3x3_tower_packed_dev (k_conv4r, csrc/sgo_conv4r.hpp: weights from a fragment-order filter bank straight into
    registers): against torch conv2d in fp32 (fp16 output rounding: 2e-3 relative + 2e-3 absolute), bit for bit against
    sgo_conv3x3_tower_dev (same MFMA order per output), bit-identical across relaunches (the race screen of its counted waits),
    exact on integer data, through the slice loop, on full / ragged / single-tile / degenerate boards; bad arguments are errors."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    torch.manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    nbytes = lib.sgo_conv3x3_tower_packed_bytes()
    assert nbytes == 4 * 36 * 8192 + 8192

    def bank_of(w):
        bank = torch.full((nbytes,), 0x5A, device="cuda", dtype=torch.uint8)
        L.check(lib.sgo_conv3x3_tower_prepack_dev(w.data_ptr(), bank.data_ptr(), st))
        return bank

    for (n, h, wd, with_skip) in [(1, 7, 7, True), (1, 17, 17, False), (3, 17, 17, True), (5, 17, 17, False), (64, 7, 7, True),
                                  (7, 5, 19, True), (2, 19, 19, False), (333, 17, 17, True), (1024, 17, 17, True),
                                  (300, 11, 1, False), (9, 1, 1, True), (40, 1, 13, True), (77, 2, 2, False), (700, 9, 9, True)]:
        x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
        w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
        b = torch.randn(256, device="cuda").half()
        skip = torch.randn(n, h, wd, 256, device="cuda").half() if with_skip else None
        sp = None if skip is None else skip.data_ptr()
        bank = bank_of(w)
        # the bank is a permutation of the filter bank's bytes (+ 8 KB of zero padding the last K-tile's look-ahead reads)
        assert torch.equal(torch.sort(bank[:nbytes - 8192].view(torch.int16)).values, torch.sort(w.view(torch.int16).reshape(-1)).values)
        assert int(bank[nbytes - 8192:].sum()) == 0
        y = torch.full((n, h, wd, 256), 7.0, device="cuda", dtype=torch.float16)
        L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), sp, y.data_ptr(), st))
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
        if with_skip:
            ref = ref + skip.float()
        ref = torch.relu(ref)
        err = (y.float() - ref).abs()
        assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (n, h, wd, with_skip, float(err.max()))
        y1 = torch.full_like(y, 5.0)
        L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), sp, y1.data_ptr(), st))
        assert torch.equal(y, y1), (n, h, wd, with_skip)
        for _ in range(3):
            y3 = torch.full_like(y, 3.0)
            L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), sp, y3.data_ptr(), st))
            assert torch.equal(y, y3)
    # the slice loop (sample cap hook) on ragged slice sizes
    x = (torch.randn(700, 17, 17, 256, device="cuda") * 0.5).half()
    skip = torch.randn_like(x)
    whole = torch.empty_like(x)
    L.check(lib.sgo_conv3x3_tower_packed_dev(700, 17, 17, x.data_ptr(), bank.data_ptr(), b.data_ptr(), skip.data_ptr(), whole.data_ptr(), st))
    for cap in (1, 255, 300):
        assert lib.sgo_conv_tower_slice_cap(cap) == 0
        try:
            part = torch.empty_like(x)
            L.check(lib.sgo_conv3x3_tower_packed_dev(700, 17, 17, x.data_ptr(), bank.data_ptr(), b.data_ptr(), skip.data_ptr(), part.data_ptr(), st))
            assert torch.equal(part, whole), cap
        finally:
            lib.sgo_conv_tower_slice_cap(0)
    del x, skip, whole, part
    # race screen at a chip-filling size: a missing wait shows up as run-to-run differences
    n, h, wd = 4096, 17, 17
    x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
    skip = torch.randn(n, h, wd, 256, device="cuda").half()
    y = torch.empty_like(skip)
    L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), st))
    y0 = y.clone()
    for _ in range(25):
        y.fill_(1.0)
        L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), st))
        assert torch.equal(y, y0)
    del x, skip, y, y0
    # exact integer data
    n, h, wd = 4, 17, 17
    x = torch.randint(-2, 3, (n, h, wd, 256), device="cuda").half()
    w = torch.randint(-1, 2, (256, 3, 3, 256), device="cuda").half()
    b = torch.randint(-3, 4, (256,), device="cuda").half()
    y = torch.empty(n, h, wd, 256, device="cuda", dtype=torch.float16)
    bank = bank_of(w)
    L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), None, y.data_ptr(), st))
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1)).permute(0, 2, 3, 1)
    assert float(ref.max()) < 2048 and torch.equal(y.float(), ref)
    # errors, not fall-backs
    assert lib.sgo_conv3x3_tower_packed_dev(1, 21, 21, x.data_ptr(), bank.data_ptr(), b.data_ptr(), None, y.data_ptr(), st) < 0
    assert lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), st) < 0
    assert lib.sgo_conv3x3_tower_prepack_dev(w.data_ptr() + 2, bank.data_ptr(), st) < 0
    assert lib.sgo_conv3x3_tower_prepack_dev(None, bank.data_ptr(), st) < 0


def test_packed_tower_route_of_the_net_equals_the_default_route(L):
    """net.FusedInferenceNet.use_packed_tower(): the whole 4-block 9x9 net (BASELINE config 2's) with its tower on k_conv4r gives
    the policy / value bits of the default route (k_conv4w)."""
    import torch
    from sejonggo_amd.net import build_fused_net
    L.load()
    net, _ = build_fused_net(9, 4, 256, name="packed_route", seed=5, device="cuda")
    X = torch.zeros(96, 9, 9, 17, device="cuda", dtype=torch.float16)
    g = torch.Generator(device="cuda").manual_seed(2)
    X[..., :16] = (torch.rand(96, 9, 9, 16, device="cuda", generator=g) < 0.2).half()
    X[..., 16] = 1.0
    p0, v0 = net.predict_on_batch(X)
    assert net.use_packed_tower(True) and len(net._banks) == 8
    p1, v1 = net.predict_on_batch(X)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)
    assert not net.use_packed_tower(False)
    p2, v2 = net.predict_on_batch(X)
    assert torch.equal(p0, p2) and torch.equal(v0, v2)


def test_tower_conv_tile_order_slices_and_bounds(L, tower_kernel):
    """The tower kernel's launch plumbing: (1) both tile orders (identity, XCD-contiguous incl. grids that are not a
    multiple of 8) give identical bits; (2) the slice loop of sgo_conv3x3_tower_dev, normally reached only beyond 2^31 bytes
    per tensor, exercised through the sample cap hook on ragged slice sizes; (3) shapes whose pixel index arithmetic would
    leave the range where the kernel's magic-number division is exact are sliced, not mis-computed (h x w = 64 x 19)."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(7)
    w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
    b = torch.randn(256, device="cuda").half()

    def conv(x, skip):
        y = torch.full_like(x, 5.0)
        L.check(lib.sgo_conv3x3_tower_dev(x.shape[0], x.shape[1], x.shape[2], x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                          skip.data_ptr(), y.data_ptr(), st))
        return y

    for n in (1, 9, 37, 300):                       # 2, 11, 42, 339 tiles: every remainder class of tiles % 8 shows up
        x = (torch.randn(n, 17, 17, 256, device="cuda") * 0.5).half()
        skip = torch.randn_like(x)
        old = lib.sgo_conv_tile_order(0)
        try:
            y0 = conv(x, skip)
            lib.sgo_conv_tile_order(1)
            y1 = conv(x, skip)
        finally:
            lib.sgo_conv_tile_order(old)
        assert torch.equal(y0, y1), n
    x = (torch.randn(700, 17, 17, 256, device="cuda") * 0.5).half()
    skip = torch.randn_like(x)
    whole = conv(x, skip)
    for cap in (1, 255, 256, 300):
        assert lib.sgo_conv_tower_slice_cap(cap) == 0
        try:
            assert torch.equal(conv(x, skip), whole), cap
        finally:
            lib.sgo_conv_tower_slice_cap(0)
    # a tall board: 64 x 19 = 1216 points per sample; (M + 256) * 1216 reaches 2^32 near 2 900 samples, far below the byte limit
    n, h, wd = 3000, 64, 19
    x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
    skip = torch.randn_like(x)
    y = conv(x, skip)
    for i in (0, 1499, 2895, 2896, 2999):
        ref = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
        ref = torch.relu(ref + skip[i:i + 1].float())
        err = (y[i:i + 1].float() - ref).abs()
        assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (i, float(err.max()))


def test_tower_conv_kernel(L, tower_kernel):
    """sgo_conv3x3_tower_dev, the hand-written CDNA4 kernel (csrc/sgo_conv8w.hpp): against torch conv2d in fp32 (tolerance:
    fp16 output rounding, 2e-3 relative + 2e-3 absolute), against the generic back end, bit-identical across repeated
    launches (the race screen for its hand-placed waits), on full / ragged / single-tile batches and both tower sizes."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    torch.manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    for (n, h, wd, with_skip) in [(1, 7, 7, True), (1, 17, 17, False), (3, 17, 17, True), (5, 17, 17, True), (5, 17, 17, False),
                                  (64, 7, 7, True), (7, 5, 19, True), (2, 19, 19, False), (333, 17, 17, True), (1024, 17, 17, True),
                                  (300, 11, 1, False), (9, 1, 1, True), (40, 1, 13, True), (77, 2, 2, False)]:   # degenerate boards: a divisor of 1
        x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
        w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
        b = torch.randn(256, device="cuda").half()
        skip = torch.randn(n, h, wd, 256, device="cuda").half() if with_skip else None
        y = torch.full((n, h, wd, 256), 7.0, device="cuda", dtype=torch.float16)
        sp = None if skip is None else skip.data_ptr()
        L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), sp, y.data_ptr(), st))
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
        if with_skip:
            ref = ref + skip.float()
        ref = torch.relu(ref)
        err = (y.float() - ref).abs()
        assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), (n, h, wd, with_skip, float(err.max()))
        # the dispatching entry point takes this shape to the same kernel: identical bits
        y1 = torch.empty_like(y)
        L.check(lib.sgo_conv3x3_bias_act_dev(n, h, wd, 256, 256, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), sp, y1.data_ptr(), st))
        assert torch.equal(y, y1)
        for _ in range(3):
            y3 = torch.full_like(y, 3.0)
            L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), sp, y3.data_ptr(), st))
            assert torch.equal(y, y3)
    # race screen at a chip-filling size: a missing wait shows up as run-to-run differences (this is how the per-wave DMA
    # count of the phase-B wait was caught)
    n, h, wd = 4096, 17, 17
    x = (torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()
    w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
    b = torch.randn(256, device="cuda").half()
    skip = torch.randn(n, h, wd, 256, device="cuda").half()
    y = torch.empty_like(skip)
    L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), st))
    y0 = y.clone()
    for _ in range(25):
        y.fill_(1.0)
        L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), st))
        assert torch.equal(y, y0)
    del x, skip, y, y0
    # exact integer data: every product and sum is representable, so the result must be exact (catches a wrong tap,
    # channel or swizzle that random data could hide inside the tolerance)
    n, h, wd = 4, 17, 17
    x = torch.randint(-2, 3, (n, h, wd, 256), device="cuda").half()
    w = torch.randint(-1, 2, (256, 3, 3, 256), device="cuda").half()
    b = torch.randint(-3, 4, (256,), device="cuda").half()
    y = torch.empty(n, h, wd, 256, device="cuda", dtype=torch.float16)
    L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), st))
    ref = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1)).permute(0, 2, 3, 1)
    assert float(ref.max()) < 2048 and torch.equal(y.float(), ref)
    # unsupported shapes are errors, not silent fall-backs
    assert lib.sgo_conv3x3_tower_dev(1, 21, 21, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), st) < 0
