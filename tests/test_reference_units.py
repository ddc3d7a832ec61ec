"""The reference's own unit tests as fixtures (tests/golden/units_S9.npz): every call that test/tests.py
TestGoMethods (:51-213), TestBoardMethods (:215-481), TestSymmetrydTestCase (:483-681), MCTSTestCase (:684-1068) and
test/tree_util_tests.py make into the path was recorded -- inputs and the value the reference returned -- while those
tests ran (and passed) against the reference in the build container (gen_golden.py child_units).  Here the same calls
go to (a) the CPU oracle and the host tree functions (no GPU) and (b) the MI355X path through the drop-in modules."""
import numpy as np
import pytest

from tests.helpers import load, unit_calls, unit_value, deser_tree, ser_tree, name_of

S = 9
SYM_K = {"left_diagonal": 1, "vertical_axis": 2, "horizontal_axis": 3, "rotation_90": 4, "rotation_180": 5, "rotation_270": 6,
         "right_diagonal": 7}


@pytest.fixture(scope="module")
def Z():
    return load("units_S9.npz")


def test_fixture_covers_the_reference_tests(Z):
    calls = unit_calls(Z)
    by = {}
    for c in calls:
        by[c["fn"]] = by.get(c["fn"], 0) + 1
    assert int(Z["tests_run"]) == 40                                  # 10 + 11 + 8 + 10 + ... all green in the reference
    assert by["capture_group"] == 17 and by["color_board"] == 6 and by["_get_points"] == 1 and by["get_liberties"] == 4
    assert by["legal_moves"] == 6 and len(Z["mp_x"]) > 400 and int(Z["n_sims"]) == 7 and int(Z["n_trees"]) == 18
    gp = [c for c in calls if c["fn"] == "_get_points"][0]
    assert unit_value(gp["out"]) == {0: 29, 1: 12, 2: 11, -1: 15, -2: 14}          # test/tests.py:135


# --------------------------------------------------------------------------------------------- CPU: oracle + host trees
def test_oracle_equals_the_reference_unit_vectors(Z):
    from oracle import oracle as ora
    for i in range(len(Z["mp_x"])):
        b = Z["mp_board_in"][i].astype(np.int32)[None]
        color = int(Z["mp_color"][i])
        _, mover = ora.make_play(int(Z["mp_x"][i]), int(Z["mp_y"][i]), b, None if color == 0 else color)
        assert np.array_equal(b[0], Z["mp_board_out"][i]) and mover == int(Z["mp_player"][i]), i
    n = 0
    for c in unit_calls(Z):
        fn, ins, want = c["fn"], [unit_value(v) for v in c["ins"]], unit_value(c["out"])
        if fn == "legal_moves":
            assert np.array_equal(ora.legal_moves(ins[0].astype(np.int32)), want), c["test"]
        elif fn == "capture_group":
            if ins[2].shape == (S, S):                     # the oracle works on S x S boards
                assert ora.capture_group(ins[0], ins[1], ins[2]) == want, c["test"]
            else:
                continue
        elif fn == "color_board":
            if ins[0].shape != (S, S):
                continue
            assert np.array_equal(ora.color_board(ins[0], ins[1]), want), c["test"]
        elif fn == "_get_points":
            assert ora.get_points(ins[0]) == want
        elif fn in SYM_K:
            assert np.array_equal(ora.sym_board(SYM_K[fn], ins[0].astype(np.int32)), want), fn
        elif fn.startswith("reverse_"):
            assert np.array_equal(ora.sym_policy_inverse(S, SYM_K[fn[8:]], ins[0].astype(np.float32)), want.astype(np.float32)), fn
        else:
            continue
        n += 1
    assert n >= 20


def _trees(Z):
    for i in range(int(Z["n_trees"])):
        p = "t%02d_" % i
        yield p, name_of(Z, p + "fn"), name_of(Z, p + "test")


def test_host_tree_functions_equal_the_reference_unit_vectors(Z):
    """tree_util.find_best_leaf_virtual_loss / get_node_by_moves, nomodel_self_play.back_propagation and
    play.tree_depth on the exact trees of test/tree_util_tests.py, compared node for node after each call."""
    from sejonggo_amd import tree_util as tu
    from sejonggo_amd.nomodel_self_play import back_propagation
    from sejonggo_amd.play import tree_depth
    seen = set()
    for p, fn, test in _trees(Z):
        tree = deser_tree(Z[p + "tin_i"], Z[p + "tin_f"])
        if fn == "find_best_leaf_virtual_loss":
            node, moves = tu.find_best_leaf_virtual_loss(tree)
            want = list(Z[p + "moves_out"])
            assert (moves if moves is not None else [-99]) == want, test
            assert (-99 if node is None else node.get("index", -98)) == int(Z[p + "leaf_index"]), test
        elif fn == "get_node_by_moves":
            if int(Z[p + "err"]):
                with pytest.raises(Exception, match="Invalid moves array"):
                    tu.get_node_by_moves(tree, list(Z[p + "moves"]))
            else:
                assert tu.get_node_by_moves(tree, list(Z[p + "moves"])).get("index", -98) == int(Z[p + "leaf_index"])
        elif fn == "back_propagation":
            leaf = deser_tree(Z[p + "leaf_i"], Z[p + "leaf_f"])
            back_propagation((leaf, [int(m) for m in Z[p + "moves"]]), tree)
        elif fn == "tree_depth":
            assert tree_depth(tree) == int(Z[p + "depth"]), test
        oi, of = ser_tree(tree)
        assert np.array_equal(oi, Z[p + "tout_i"]) and np.array_equal(of, Z[p + "tout_f"]), (fn, test)
        seen.add(fn)
    assert seen == {"find_best_leaf_virtual_loss", "get_node_by_moves", "back_propagation", "tree_depth"}


# --------------------------------------------------------------------------------------------- GPU: the drop-in modules
@pytest.fixture(scope="module")
def P():
    from sejonggo_amd import _lib, play
    _lib.require_gpu()
    return play


@pytest.mark.gpu
def test_make_play_sequences_of_the_reference_tests(Z, P):
    """All 459 make_play calls of the reference's test set-ups (suicide :250-267, captures, ko shapes, explicit colours,
    passes), one batched launch: boards and returned players bit-equal."""
    n = len(Z["mp_x"])
    boards = np.ascontiguousarray(Z["mp_board_in"].astype(np.int32))
    cols = Z["mp_color"].astype(np.int32)
    out, movers = P.make_play(Z["mp_x"].astype(np.int32), Z["mp_y"].astype(np.int32), boards, cols)
    assert np.array_equal(out, Z["mp_board_out"].astype(np.int32))
    assert np.array_equal(movers, Z["mp_player"].astype(np.int32))
    # and one by one through the scalar call form, colour None where the reference passed none
    for i in range(0, n, 23):
        b = Z["mp_board_in"][i].astype(np.int32)[None].copy()
        c = int(Z["mp_color"][i])
        b2, mover = P.make_play(int(Z["mp_x"][i]), int(Z["mp_y"][i]), b, None if c == 0 else c)
        assert b2 is b and np.array_equal(b[0], Z["mp_board_out"][i]) and mover == int(Z["mp_player"][i])


@pytest.mark.gpu
def test_go_methods_and_board_methods(Z, P):
    from sejonggo_amd import symmetry as sy
    from sejonggo_amd.conf import conf
    old = conf['SIZE']
    conf['SIZE'] = S
    seen = {}
    try:
        for c in unit_calls(Z):
            fn, ins, want = c["fn"], [unit_value(v) for v in c["ins"]], unit_value(c["out"])
            if fn == "legal_moves":
                got = P.legal_moves(ins[0].astype(np.int32))
                assert got.dtype == np.int64 and np.array_equal(got, want), c["test"]
            elif fn == "capture_group":
                assert P.capture_group(ins[0], ins[1], ins[2]) == want, (c["test"], ins[0], ins[1])   # order included
            elif fn == "color_board":
                src = ins[0].copy()
                got = P.color_board(ins[0], ins[1])
                assert np.array_equal(got, want) and np.array_equal(ins[0], src), c["test"]
            elif fn == "_get_points":
                assert P._get_points(ins[0]) == want
            elif fn == "get_liberties":
                got = P.get_liberties(ins[0], ins[1], ins[2].astype(np.int32), ins[3])
                assert sorted(got) == want, c["test"]
            elif fn in SYM_K:
                b = ins[0].astype(np.int32)
                got = getattr(sy, fn)(b)
                assert np.array_equal(got, want), fn
                if fn in ("vertical_axis", "horizontal_axis"):
                    assert got is b                                       # in place, symmetry.py:54-56,77-79
            elif fn.startswith("reverse_"):
                pol = ins[0].astype(np.float32)
                got = getattr(sy, fn)(pol)
                assert np.array_equal(got, want.astype(np.float32)), fn
            else:
                raise AssertionError("unhandled recorded call " + fn)
            seen[fn] = seen.get(fn, 0) + 1
    finally:
        conf['SIZE'] = old
    assert len(seen) == 19


class _ReplayModel(object):
    """Returns the recorded network outputs and checks that the boards the reference's model saw arrive."""
    name = "replay"

    def __init__(self, Z, p):
        self.calls = [(Z[p + "X%d" % j], Z[p + "P%d" % j], Z[p + "V%d" % j]) for j in range(int(Z[p + "n_pred"]))]
        self.i = 0

    def predict_on_batch(self, X):
        want, pol, val = self.calls[self.i]
        self.i += 1
        assert np.array_equal(np.asarray(X).astype(np.int8), want), "the model was fed different boards than in the reference"
        return pol.copy(), val.copy()


@pytest.mark.gpu
def test_simulate_on_the_trees_of_the_reference_mcts_tests(Z, P):
    """self_play.simulate on the hand-built trees of MCTSTestCase (tests.py:731-1068): same boards reach the model,
    same tree afterwards (counts, values, means, priors of the new children), same board mutation."""
    from sejonggo_amd import self_play as sp
    from sejonggo_amd import symmetry as sy
    from sejonggo_amd.conf import conf
    old, old_sym = conf['SIZE'], sy.SYMMETRIES
    conf['SIZE'] = S
    sy.SYMMETRIES = sy.SYMMETRIES[0:1]                                    # tests.py:688-689
    try:
        for i in range(int(Z["n_sims"])):
            p = "s%02d_" % i
            tree = deser_tree(Z[p + "tin_i"], Z[p + "tin_f"])
            board = Z[p + "board_in"].astype(np.int32)
            model = _ReplayModel(Z, p)
            sp.simulate(tree, board, model, int(Z[p + "batch"]), int(Z[p + "orig"]))
            assert model.i == len(model.calls), name_of(Z, p + "test")
            oi, of = ser_tree(tree)
            assert np.array_equal(oi[:, :7], Z[p + "tout_i"][:, :7]), name_of(Z, p + "test")
            assert np.array_equal(of, Z[p + "tout_f"]), name_of(Z, p + "test")
            assert np.array_equal(oi[:, 7:], Z[p + "tout_i"][:, 7:]), "scalar types (float32 / float64 regime) differ"
            assert np.array_equal(board, Z[p + "board_out"].astype(np.int32))
    finally:
        conf['SIZE'], sy.SYMMETRIES = old, old_sym
