"""HDF5 plumbing of the path's two file formats, checked against the HDF5 C library itself (libhdf5, the library h5py
wraps; found under /opt/conda/lib in this image -- h5py is not installed):

* sample.h5 (sgfsave.py:49-79): what the pure-Python writer hdf5_min.py emits is read back, value for value, by libhdf5
  (through sejonggo_amd/h5lite.py and, where present, the h5dump tool); sgfsave writes through libhdf5 when it can.
* Keras model files (model.py:147-157): a file in Keras' layout written with libhdf5 is found and loaded by the model
  loaders (name from `model_config`, weights from `model_weights/<layer>/<layer>/<weight>:0`), bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from sejonggo_amd import h5lite

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="the HDF5 C library (libhdf5) is not present")


def test_library_version():
    v = h5lite.libversion()
    assert v[0] == 1 and v[1] >= 8


def test_hdf5_min_files_are_read_by_libhdf5(tmp_path):
    """The spec-following writer against the real reader: every shape the path writes (19x19 and 9x9 boards, the scalar
    value target), plus awkward ones (empty-ish, long 1-D, 5-D)."""
    from sejonggo_amd.hdf5_min import write_datasets
    rng = np.random.RandomState(0)
    cases = [
        {"board": rng.randint(-1, 2, (1, 19, 19, 17)).astype(np.float32), "policy_target": rng.rand(362).astype(np.float32),
         "value_target": np.array(-1.0, dtype=np.float32)},
        {"board": rng.randint(0, 2, (1, 9, 9, 17)).astype(np.float32), "policy_target": rng.rand(82).astype(np.float32),
         "value_target": np.array(1.0, dtype=np.float32)},
        {"a": np.arange(3, dtype=np.float32), "bb": rng.rand(2, 3, 4, 5, 6).astype(np.float32), "c": np.array(0.5, dtype=np.float32),
         "a_rather_long_dataset_name_to_move_the_link_messages": rng.rand(1000).astype(np.float32)},
    ]
    for i, data in enumerate(cases):
        p = str(tmp_path / ("f%d.h5" % i))
        write_datasets(p, data)
        with h5lite.File(p) as f:
            assert sorted(f.keys()) == sorted(data)
            for k, want in data.items():
                got = f[k]
                assert got.shape == want.shape and got.dtype == np.float32
                assert np.array_equal(got[...] if want.shape else got[()], want), (i, k)
    tool = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(tool):
        out = subprocess.run([tool, "-d", "/value_target", str(tmp_path / "f0.h5")], capture_output=True, text=True)
        assert out.returncode == 0 and "H5T_IEEE_F32LE" in out.stdout and "(0): -1" in out.stdout


def test_sgfsave_writes_through_libhdf5_and_train_side_reads(tmp_path):
    """save_self_play_data -> the three datasets train.py:113-119 reads (board, policy_target, value_target), through libhdf5."""
    from sejonggo_amd import sgfsave
    from sejonggo_amd.conf import conf
    old = dict(conf)
    conf.update(SELF_PLAY_DIR=str(tmp_path), SIZE=9)
    try:
        S = 9
        rng = np.random.RandomState(1)
        moves = [{'board': rng.randint(0, 2, (1, S, S, 17)).astype(np.int32), 'policy': rng.rand(S * S + 1), 'player': 1 if k % 2 == 0 else -1,
                  'move_n': k, 'value': np.float32(0.1)} for k in range(3)]
        sgfsave.save_self_play_data("model_7", 12, {'winner': 1, 'moves': moves})
        for k, mv in enumerate(moves):
            p = os.path.join(str(tmp_path), "model_7", "game_00012", "move_%03d" % k, "sample.h5")
            with h5lite.File(p) as f:
                assert np.array_equal(f['board'][...], mv['board'].astype(np.float32))
                assert np.array_equal(f['policy_target'][...], mv['policy'].astype(np.float32))
                assert f['value_target'].shape == () and f['value_target'][()] == (1.0 if mv['player'] == 1 else -1.0)   # sgfsave.py:56
        if not sgfsave.HAVE_H5:
            # libhdf5 wrote the file itself: default library format (superblock 0), not the hdf5_min one
            raw = open(p, "rb").read(9)
            assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0
    finally:
        conf.clear()
        conf.update(old)


def test_writer_threads_share_the_library_safely(tmp_path):
    """run_selfplay hands finished games to writer THREADS; libhdf5 is not thread-safe, so sgfsave serialises on h5lite.LOCK."""
    from concurrent.futures import ThreadPoolExecutor
    from sejonggo_amd import sgfsave
    from sejonggo_amd.conf import conf
    old = dict(conf)
    conf.update(SELF_PLAY_DIR=str(tmp_path), SIZE=9)
    try:
        S = 9

        def game(g):
            rng = np.random.RandomState(g)
            moves = [{'board': rng.randint(0, 2, (1, S, S, 17)).astype(np.int32), 'policy': rng.rand(S * S + 1), 'player': 1,
                      'move_n': k, 'value': np.float32(0)} for k in range(12)]
            sgfsave.save_self_play_data("m", g, {'winner': 1, 'moves': moves})
            return moves

        with ThreadPoolExecutor(max_workers=4) as ex:
            all_moves = list(ex.map(game, range(16)))
        for g, moves in enumerate(all_moves):
            for k in (0, 11):
                with h5lite.open(os.path.join(str(tmp_path), "m", "game_%05d" % g, "move_%03d" % k, "sample.h5")) as f:
                    assert np.array_equal(f['board'][...], moves[k]['board'].astype(np.float32))
    finally:
        conf.clear()
        conf.update(old)


def test_attribute_forms(tmp_path):
    p = str(tmp_path / "a.h5")
    with h5lite.File(p, "w") as f:
        f.attrs["fixed"] = b"tensorflow"
        f.attrs["vlen"] = "variable-length ✓"
        f.attrs["names"] = np.array([b"conv2d_1", b"batch_normalization_12", b"x"], dtype="S")
        f.attrs["n"] = np.int64(7)
        f.attrs["v"] = np.array([1.5, 2.5], dtype=np.float32)
        g = f.create_group("grp")
        g.create_dataset("deep/er/data:0", data=np.arange(6, dtype=np.float32).reshape(2, 3))
    with h5lite.File(p) as f:
        assert f.attrs["fixed"] == b"tensorflow" and f.attrs["vlen"] == "variable-length ✓".encode("utf8")
        assert list(f.attrs["names"]) == [b"conv2d_1", b"batch_normalization_12", b"x"]
        assert f.attrs["n"] == 7 and list(f.attrs["v"]) == [1.5, 2.5]
        assert "missing" not in f.attrs and f.attrs.get("missing", 3) == 3
        assert "grp/deep/er/data:0" in f and "grp/nope/x" not in f
        assert np.array_equal(f["grp"]["deep/er/data:0"][...], np.arange(6, dtype=np.float32).reshape(2, 3))
        with pytest.raises(KeyError):
            f["nope"]
    with pytest.raises(OSError):
        h5lite.File(str(tmp_path / "absent.h5"))


def test_keras_model_file_round_trip_through_the_loaders(tmp_path):
    """model.load_best_model / load_latest_model / model_name on Keras-layout .h5 files (keras/engine/saving.py: root attrs
    model_config / keras_version / backend, group model_weights, attrs layer_names / weight_names, datasets
    <layer>/<weight>:0 in Keras array layouts) written with libhdf5."""
    import torch
    from sejonggo_amd import keras_import as ki, model as M
    from sejonggo_amd.conf import conf
    from sejonggo_amd.net import PolicyValueNet
    old = dict(conf)
    conf.update(MODEL_DIR=str(tmp_path), SIZE=9, N_RESIDUAL_BLOCKS=2, NET_CHANNELS=16)
    try:
        torch.manual_seed(4)
        net = PolicyValueNet(9, 2, 16, name="model_5")
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        ki.save_keras_h5(os.path.join(str(tmp_path), "best_model.h5"), net)
        ki.save_keras_h5(os.path.join(str(tmp_path), "model_5.h5"), net)
        assert ki.keras_model_name(os.path.join(str(tmp_path), "best_model.h5")) == "model_5"
        assert M.model_name("BEST") == "model_5" and M.model_name("LATEST") == "model_5"
        with h5lite.File(os.path.join(str(tmp_path), "best_model.h5")) as f:
            g = f["model_weights"]
            names = [n.decode() for n in g.attrs["layer_names"]]
            assert names[:2] == ["conv2d_1", "batch_normalization_1"] and "policy_out" in names and "value_out" in names
            k = g["conv2d_1"]["conv2d_1/kernel:0"]
            assert k.shape == (3, 3, 17, 16)                                      # Keras Conv2D kernel layout [kh][kw][in][out]
            assert [n.decode() for n in g["batch_normalization_1"].attrs["weight_names"]] == [
                "batch_normalization_1/gamma:0", "batch_normalization_1/beta:0", "batch_normalization_1/moving_mean:0",
                "batch_normalization_1/moving_variance:0"]
        for loaded in (M.load_best_model(), M.load_latest_model(), M.load_model_by_name("model_5.h5")):
            assert loaded.name == "model_5"
            for (ka, a), (kb, b) in zip(net.state_dict().items(), loaded.state_dict().items()):
                assert ka == kb and (a.dtype != torch.float32 or torch.equal(a, b)), ka
        x = torch.zeros(2, 9, 9, 17)
        x[..., 16] = 1
        p0, v0 = net.eval().predict_on_batch(x)
        p1, v1 = M.load_best_model().predict_on_batch(x)
        assert torch.equal(p0, p1) and torch.equal(v0, v1)
    finally:
        conf.clear()
        conf.update(old)
