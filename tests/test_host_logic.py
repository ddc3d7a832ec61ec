"""CPU-only tests of the host-side logic around the HIP path: game scheduling / resign calibration
(selfplay_worker.py:82-118), sample writer rules (sgfsave.py:49-79), record unpacking, stub nets."""
import os

import numpy as np

from tests.helpers import load, read_sample


def test_game_scheduler_reserves_directories(tmp_path):
    from sejonggo_amd.selfplay_worker import GameScheduler
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "m", "game_00001"))       # someone else's game
    s = GameScheduler(root, "m", 4, 0.1, 0.05, rand=lambda: 0.5)
    got = [s.reserve() for _ in range(5)]
    assert got == [0, 2, 3, None, None]
    assert os.path.isdir(os.path.join(root, "m", "game_00003"))
    s.discard(3)
    assert not os.path.isdir(os.path.join(root, "m", "game_00003"))


def test_resign_threshold_matches_reference_rule():
    """selfplay_worker.py:100-112: arrival-order list, index int(0.05 * len), winner's values only."""
    from sejonggo_amd.selfplay_worker import GameScheduler
    s = GameScheduler("/nonexistent", "m", 0, 0.10, 0.05, rand=lambda: 0.5)
    assert s.pick_resign() is None
    rng = np.random.RandomState(0)
    mins = []
    for g in range(45):
        moves = [{'value': np.float32(v)} for v in rng.uniform(-1, 1, size=10)]
        winner = 1 if g % 2 == 0 else 0
        s.finished({'moves': moves, 'winner': winner}, None)
        vals = [m['value'] for m in (moves[::2] if winner == 1 else moves[1::2])]
        mins.append(min(vals))
        idx = int(0.05 * len(mins))
        want = mins[idx] if idx > 0 else None
        if idx > 0:
            assert s.current_resign == want
    assert s.pick_resign() == s.current_resign
    s.rand = lambda: 0.05            # 10 % of games play without resignation
    assert s.pick_resign() is None
    before = list(s.min_values)
    s.finished({'moves': [{'value': np.float32(-1)}] * 4, 'winner': 1}, resign=-0.9)   # resign games do not calibrate
    assert s.min_values == before


def test_value_target_compat_and_corrected():
    from sejonggo_amd.sgfsave import value_target
    # reference rule (sgfsave.py:56): winner in {1, 0, None}, player in {+1, -1}
    assert value_target(1, 1, 0, compat=True) == 1 and value_target(1, -1, 1, compat=True) == -1
    assert value_target(0, 1, 0, compat=True) == -1 and value_target(0, -1, 1, compat=True) == -1   # white wins: all -1
    assert value_target(None, 1, 0, compat=True) == -1
    assert value_target(0, 0, 1, compat=False) == 1 and value_target(0, 0, 2, compat=False) == -1
    assert value_target(None, 0, 3, compat=False) == 0


def test_sample_writer_layout(tmp_path):
    from sejonggo_amd import sgfsave
    from sejonggo_amd.conf import conf
    old = conf['SELF_PLAY_DIR']
    conf['SELF_PLAY_DIR'] = str(tmp_path)
    try:
        S = 9
        gd = {'winner': 1, 'moves': [{'board': np.ones((1, S, S, 17), np.int32), 'policy': np.full(S * S + 1, 0.5), 'player': 1,
                                      'move_n': k, 'value': np.float32(0.1)} for k in range(3)]}
        sgfsave.save_self_play_data("model_0", 7, gd)
        d = os.path.join(str(tmp_path), "model_0", "game_00007", "move_002")
        assert os.path.isdir(d)
        assert os.path.isfile(os.path.join(d, "sample.h5"))
        b, p, v = read_sample(os.path.join(d, "sample.h5"))
        assert b.shape == (1, S, S, 17) and b.dtype == np.float32
        assert p.shape == (S * S + 1,) and p.dtype == np.float32 and v.dtype == np.float32 and v.shape == () and v == 1
    finally:
        conf['SELF_PLAY_DIR'] = old


def test_unpack_positions_inverts_the_packed_layout():
    """Packed record (include/sgo.h): plane 2k = black / 2k+1 = white k plies ago, to-play bit = top bit of plane 0's
    last word; the board tensor's planes are relative to the side to move."""
    from sejonggo_amd.engine import unpack_positions
    rng = np.random.RandomState(1)
    for S in (5, 9, 19):
        N = S * S
        NW = (N + 31) // 32
        RW = 16 * NW
        boards = np.zeros((6, S, S, 17), np.int32)
        boards[..., :16] = rng.randint(0, 2, size=(6, S, S, 16))
        boards[..., 16] = rng.choice([-1, 1], size=6)[:, None, None]
        packed = np.zeros((6, RW), np.uint32)
        for b in range(6):
            white = boards[b, 0, 0, 16] == -1
            for c in range(16):
                absolute = c ^ 1 if white else c
                for a in np.flatnonzero(boards[b, :, :, c].reshape(-1)):
                    packed[b, absolute * NW + (a >> 5)] |= np.uint32(1 << (a & 31))
            if white:
                packed[b, NW - 1] |= np.uint32(1 << 31)
        assert np.array_equal(unpack_positions(packed, S), boards)


def test_stub_nets_numpy_equals_torch():
    import torch
    from sejonggo_amd.stub_nets import make_stub
    rng = np.random.RandomState(2)
    for kind in ("uniform", "dummy", "hash"):
        net = make_stub(kind, 9)
        X = rng.randint(0, 2, size=(5, 9, 9, 17)).astype(np.int32)
        X[..., 16] = -1
        p1, v1 = net.predict_on_batch(X)
        p2, v2 = net.predict_on_batch(torch.from_numpy(X).to(torch.float16))
        assert p1.dtype == np.float32 and p1.tobytes() == p2.numpy().tobytes()
        assert v1.tobytes() == v2.numpy().tobytes()


def test_dummy_net_restates_reference_dummy_model():
    """DummyModel (test/tests.py:34-49): policy proportional to reversed(range(1, A+1)), value 1."""
    from sejonggo_amd.stub_nets import DummyNet
    p, v = DummyNet(9).predict_on_batch(np.zeros((2, 9, 9, 17), np.float32))
    want = np.array(list(reversed(range(1, 83))), dtype=np.float32)
    want /= want.sum()
    assert np.allclose(p[0], want, rtol=0, atol=1e-9) and (v == 1).all()
    z = load("async_01.npz")   # the golden game played with it has value == 1 at every root
    assert (z["move_value"] == 1).all()


def test_net_contract_and_bn_folding():
    import torch
    from sejonggo_amd.net import PolicyValueNet
    torch.manual_seed(0)
    net = PolicyValueNet(9, 2, 16).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2.0)
    X = torch.randint(-1, 2, (3, 9, 9, 17)).float()
    p, v = net.predict_on_batch(X)
    assert p.shape == (3, 82) and v.shape == (3, 1) and torch.allclose(p.sum(1), torch.ones(3), atol=1e-5)
    assert (v.abs() <= 1).all()
    p2, v2 = net.fused(torch.float32).predict_on_batch(X)
    assert float((p - p2).abs().max()) < 1e-6 and float((v - v2).abs().max()) < 1e-5
    # reference topology: 'valid' stem => tower is (S-2)^2 ; FLOPs as SURVEY.md §8d counts them
    assert PolicyValueNet(19, 20, 256).tower_side == 17
    assert abs(PolicyValueNet(19, 20, 256).flops_per_eval() / 1e9 - 13.66) < 0.01


def test_hdf5_min_checksum_and_round_trip(tmp_path):
    """lookup3 known answers (lookup3.c driver: "Four score and seven years ago") and a write/read round trip; when
    h5py is importable the spec-written file must also open through libhdf5."""
    from sejonggo_amd.hdf5_min import lookup3, write_datasets, read_datasets
    s = b"Four score and seven years ago"
    assert (lookup3(b""), lookup3(s, 0), lookup3(s, 1)) == (0xdeadbeef, 0x17770551, 0xcd628161)
    rng = np.random.RandomState(0)
    want = {'board': rng.randint(-1, 2, size=(1, 19, 19, 17)).astype(np.float32),
            'policy_target': rng.rand(362).astype(np.float32), 'value_target': np.array(-1, dtype=np.float32)}
    path = str(tmp_path / "sample.h5")
    write_datasets(path, want)
    got = read_datasets(path)
    assert list(got) == list(want)
    for k in want:
        assert got[k].shape == want[k].shape and got[k].dtype == np.float32 and np.array_equal(got[k], want[k])
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    try:
        import h5py
    except Exception:
        return
    with h5py.File(path, "r") as f:
        for k in want:
            assert np.array_equal(f[k][()], want[k])


def test_keras_layer_mapping_round_trip():
    """keras_import: a PolicyValueNet exported to Keras-layout arrays (model.py:55-95 layer order and names) and read back
    from a shuffled layer list is the same network bit for bit; wrong depth and wrong shapes are errors.  (The HDF5 container
    itself needs h5py: parity unpinned, no Keras file ships with the reference.)"""
    import random
    import numpy as np
    import pytest
    import torch
    from sejonggo_amd.keras_import import assign_keras_layers, export_keras_layers
    from sejonggo_amd.net import PolicyValueNet
    torch.manual_seed(3)
    a = PolicyValueNet(9, n_blocks=5, channels=16, name="a").eval()
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    layers = export_keras_layers(a, first_index=7)          # conv2d_7 .. conv2d_19: numeric, not lexicographic, order
    assert layers[0][1][0].shape == (3, 3, 17, 16) and layers[-1][1][0].shape == (256, 1)
    layers += [("activation_3", []), ("add_1", [])]
    random.Random(0).shuffle(layers)
    b = PolicyValueNet(9, n_blocks=5, channels=16, name="b").eval()
    assign_keras_layers(b, layers)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(sa[k], sb[k]), k
    x = (torch.rand(3, 9, 9, 17) < 0.2).float()
    pa, va = a.predict_on_batch(x)
    pb, vb = b.predict_on_batch(x)
    assert torch.equal(pa, pb) and torch.equal(va, vb)
    with pytest.raises(ValueError):
        assign_keras_layers(PolicyValueNet(9, n_blocks=4, channels=16), layers)
    with pytest.raises(ValueError):
        assign_keras_layers(PolicyValueNet(9, n_blocks=5, channels=8), layers)
    with pytest.raises(ValueError):
        assign_keras_layers(PolicyValueNet(9, n_blocks=5, channels=16), [l for l in layers if l[0] != "policy_out"])


def test_symmetry_index_builders_reproduce_the_library_tables():
    """symmetry.py:12-42: rotation_indexes / axis_symmetry_indexes (float rotation + round) at the angles the reference uses
    give exactly the SWAP tables of the library (which the goldens pin), and the module-level *_SWAP names resolve."""
    import math
    from sejonggo_amd import symmetry as sy
    from sejonggo_amd.conf import conf
    old = conf['SIZE']
    try:
        for S in (5, 9, 19):
            conf['SIZE'] = S
            want = {k: [int(v) for v in sy.sym_lut(S, k)] for k in range(8)}
            assert sy.axis_symmetry_indexes(math.pi / 4) == want[1] == sy.LEFT_DIAGONAL_SWAP
            assert sy.axis_symmetry_indexes(math.pi / 2) == want[2] == sy.VERTICAL_AXIS_SWAP
            assert sy.axis_symmetry_indexes(0) == want[3] == sy.HORIZONTAL_AXIS_SWAP
            assert sy.rotation_indexes(math.pi / 2) == want[4] == sy.ROTATION_90_SWAP
            assert sy.rotation_indexes(math.pi) == want[5] == sy.ROTATION_180_SWAP
            assert sy.rotation_indexes(3 * math.pi / 2) == want[6] == sy.ROTATION_270_SWAP
            assert sy.axis_symmetry_indexes(-math.pi / 4) == want[7] == sy.RIGHT_DIAGONAL_SWAP
    finally:
        conf['SIZE'] = old


def test_self_play_dir_statistics_and_clean_up(tmp_path):
    """sgfsave.py:83-127."""
    from sejonggo_amd import sgfsave
    for model, games in (("m1", {0: 3, 1: 60, 2: 0}), ("m2", {5: 10})):
        for g, n in games.items():
            for k in range(n):
                os.makedirs(os.path.join(str(tmp_path), model, "game_%05d" % g, "move_%03d" % k))
            os.makedirs(os.path.join(str(tmp_path), model, "game_%05d" % g), exist_ok=True)
    st = sgfsave.statistic_by_model(os.path.join(str(tmp_path), "m1"), 50)
    assert st[3] == ["game_00000"] and st[0] == ["game_00002"] and sum(len(v) for v in st.values()) == 2
    al = sgfsave.statistic_all_model(str(tmp_path), 50)
    assert al[10] == ["game_00005"] and sum(len(v) for v in al.values()) == 3
    assert sgfsave.clean_up(str(tmp_path), 50) == 3
    assert os.listdir(os.path.join(str(tmp_path), "m1")) == ["game_00001"] and os.listdir(os.path.join(str(tmp_path), "m2")) == []


def test_reference_names_are_all_present():
    """Every public function / class of the reference modules ON THE HOT PATH has a counterpart of the same
    name in the mirror.  Not mirrored on purpose: the reference's superseded root-parallel search (async_simulate / update_root
    / basic_tasks) and its debug printers (show_tree, show_board_old, _color_adjoint) -- no §8 row names them."""
    import importlib
    want = {
        "play": ["str_coord", "index2coord", "coord2index", "gtpcoord2index", "get_surrounding", "get_liberties", "get_real_board",
                 "_show_board", "show_board", "capture_group", "take_stones", "swap_player", "make_play",
                 "color_board", "get_winner", "_get_points", "game_init", "choose_first_player",
                 "top_one_with_virtual_loss", "top_one_action", "top_n_actions", "tree_depth", "new_tree", "new_subtree",
                 "legal_moves"],
        "symmetry": ["_id", "rotation_indexes", "axis_symmetry_indexes", "left_diagonal", "reverse_left_diagonal", "right_diagonal",
                     "reverse_right_diagonal", "vertical_axis", "reverse_vertical_axis", "horizontal_axis", "reverse_horizontal_axis",
                     "rotation_90", "reverse_rotation_90", "rotation_180", "reverse_rotation_180", "rotation_270",
                     "reverse_rotation_270", "random_symmetry_predict", "SYMMETRIES"],
        "tree_util": ["find_best_leaf_virtual_loss", "get_node_by_moves"],
        "nomodel_self_play": ["back_propagation", "async_simulate2", "select_play",
                              "play_game_async"],
        "self_play": ["simulate", "mcts_decision", "select_play", "play_game", "model_self_play", "self_play"],
        "simulation_workers": ["init_simulation_workers", "init_simulation_workers_by_gpuid", "init_pool_param",
                               "destroy_simulation_workers", "basic_tasks2", "board_worker", "subtree_worker",
                               "simulation_result_queue", "process_pool"],
        "predicting_queue_worker": ["init_predicting_workers", "destroy_predicting_workers", "PredictingQueueWorker",
                                    "put_name_request", "put_predict_request"],
        "selfplay_worker": ["SelfPlayWorker", "NoModelSelfPlayWorker"],
        "evaluate_worker": ["NoModelEvaluateWorker"],
        "main_selfplay": ["main"],
        "sgfsave": ["save_file", "save_game_data", "save_self_play_data", "clean_up", "statistic_all_model", "statistic_by_model",
                    "save_game_sgf"],
        "go_game": ["GoGame", "IllegalMove", "WHITE", "BLACK", "EMPTY", "RESIGN", "PASS"],
        "model": ["build_model", "create_initial_model", "load_latest_model", "load_best_model", "load_model_by_name"],
        "evaluator": ["elect_model_as_best_model", "evaluate", "eval_statistic", "promote_best_model", "clean_up_result"],
        "utils": ["init_directories", "clean_up_empty", "prRed", "prGreen", "prYellow", "prLightPurple", "prPurple", "prCyan",
                  "prLightGray", "prBlack"],
    }
    for mod, names in want.items():
        m = importlib.import_module("sejonggo_amd." + mod)
        for n in names:
            assert hasattr(m, n), "%s.%s" % (mod, n)
    from sejonggo_amd import predicting_queue_worker as pq
    w = pq.PredictingQueueWorker(3)
    assert w.gpu_id == 3 and w.join() is None
