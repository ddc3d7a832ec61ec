"""The self-play ENTRY path: main_selfplay.main() -> init_predicting_workers / put_name_request in a GPU-free
parent -> NoModelSelfPlayWorker processes (main_selfplay.py:9-29, selfplay_worker.py:61-130).

CPU part: the parent never touches the GPU runtime; the reference's own main_selfplay.py runs unchanged on this
package's modules (INTEGRATION.md §A recipe) and starts / joins N_GAME_PROCESS workers; model loader branches.
GPU part: the whole flow on the device, by fork (fresh parent) and by spawn (parent that already used the GPU),
the sync SelfPlayWorker, and that the shipped path runs the hand-written tower convolution."""
import glob
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

CONF_SNIPPET = """
import os, sys
sys.path.insert(0, {root!r})
from sejonggo_amd.conf import conf
conf.update(MODEL_DIR={tmp!r} + '/models', SELF_PLAY_DIR={tmp!r} + '/selfplay', EVAL_DIR={tmp!r} + '/eval',
            LOG_DIR={tmp!r} + '/logs', TMP_DIR={tmp!r} + '/tmp', GAMES_DIR={tmp!r} + '/eval')
"""


def _run(script, timeout=600):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=timeout, env=env, cwd="/tmp")
    assert r.returncode == 0, "child failed:\nSTDOUT:\n%s\nSTDERR:\n%s" % (r.stdout[-4000:], r.stderr[-4000:])
    return r.stdout


# ------------------------------------------------------------------------------------------------- CPU
def test_parent_side_calls_do_not_touch_the_gpu(tmp_path):
    """init_predicting_workers + put_name_request("BEST") -- what main() runs before it forks -- must neither load
    the HIP library nor initialise torch.cuda, and report the initial model's name (model.py:155)."""
    out = _run(CONF_SNIPPET.format(root=ROOT, tmp=str(tmp_path)) + textwrap.dedent("""
        from sejonggo_amd import _lib
        from sejonggo_amd.predicting_queue_worker import init_predicting_workers, put_name_request, destroy_predicting_workers
        init_predicting_workers(conf['GPUs'])
        name = put_name_request("BEST")
        assert put_name_request("BEST_NAME") == put_name_request("BEST_SYM") == name
        import torch
        assert _lib._lib is None, "libsgo_hip.so was loaded in the parent"
        assert _lib.gpu_touched_pid is None and not _lib.gpu_runtime_initialised()
        assert not torch.cuda.is_initialized()
        destroy_predicting_workers(conf['GPUs'])
        print("NAME", name)
    """))
    assert "NAME model_1" in out


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "main_selfplay.py")), reason="reference checkout absent")
def test_reference_main_selfplay_runs_on_the_mirrors(tmp_path):
    """the reference's main_selfplay.py, unchanged, with this package's modules under the
    reference's names.  run_selfplay (the GPU body) is replaced by a marker writer: the test is about imports,
    the loop, and that N_GAME_PROCESS workers are started (by fork, from a GPU-free parent) and joined."""
    out = _run(CONF_SNIPPET.format(root=ROOT, tmp=str(tmp_path)) + textwrap.dedent("""
        import importlib
        for name in ("conf", "play", "symmetry", "tree_util", "self_play", "nomodel_self_play", "predicting_queue_worker",
                     "simulation_workers", "selfplay_worker", "sgfsave", "utils", "model", "go_game"):
            sys.modules[name] = importlib.import_module("sejonggo_amd." + name)
        conf['N_GAME_PROCESS'] = 2
        conf['GPUs'] = [0, 1]
        import selfplay_worker
        from sejonggo_amd import _lib

        def fake_run_selfplay(gpu_id, indicator, **kw):
            assert not _lib.gpu_runtime_initialised(), "worker inherited an initialised GPU runtime"
            open(os.path.join(%r, "worker_gpu%%d_pid%%d_ppid%%d" %% (gpu_id, os.getpid(), os.getppid())), "w").write(indicator)
            return 0
        selfplay_worker.run_selfplay = fake_run_selfplay
        sys.path.append(%r)
        import main_selfplay
        assert main_selfplay.__file__.startswith(%r)
        assert main_selfplay.NoModelSelfPlayWorker is selfplay_worker.NoModelSelfPlayWorker
        main_selfplay.main()
        print("PARENT", os.getpid())
    """ % (str(tmp_path), REFERENCE, REFERENCE)))
    assert "SELF-PLAYING BEST MODEL  model_1" in out
    assert "No new best model for self-playing. Stopping.." in out
    parent = int(out.split("PARENT")[1].split()[0])
    marks = sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "worker_gpu*")))
    assert len(marks) == 2 and marks[0].startswith("worker_gpu0_") and marks[1].startswith("worker_gpu1_")
    assert all(m.endswith("_ppid%d" % parent) for m in marks)         # children of main(), one per process id
    assert open(glob.glob(os.path.join(str(tmp_path), "worker_gpu0*"))[0]).read() == "BEST_SYM"
    assert os.path.isdir(os.path.join(str(tmp_path), "selfplay")) and os.path.isdir(os.path.join(str(tmp_path), "models"))


def _with_conf(tmp_path, **kw):
    from sejonggo_amd.conf import conf
    old = dict(conf)
    t = str(tmp_path)
    conf.update(MODEL_DIR=t + '/models', SELF_PLAY_DIR=t + '/selfplay', EVAL_DIR=t + '/eval', LOG_DIR=t + '/logs',
                TMP_DIR=t + '/tmp', GAMES_DIR=t + '/eval')
    conf.update(kw)
    return conf, old


def _restore(conf, old):
    conf.clear()
    conf.update(old)


def test_model_loader_branches(tmp_path):
    """model.py:125-163 mirror: .pt round trip by name, latest = highest _<n>, a Keras .h5 without h5py is an error
    (never a silent random-init), no file at all = the initial model `model_1`, saved like create_initial_model."""
    import torch
    from sejonggo_amd import model as M
    conf, old = _with_conf(tmp_path, SIZE=5, N_RESIDUAL_BLOCKS=1, NET_CHANNELS=8)
    try:
        assert M.model_name("BEST") == "model_1" and M.model_name("LATEST") == "model_1"
        net = M.load_best_model()                                  # creates model_1 + best_model, loudly
        assert net.name == "model_1"
        assert os.path.isfile(os.path.join(conf['MODEL_DIR'], "model_1.pt"))
        assert os.path.isfile(os.path.join(conf['MODEL_DIR'], "best_model.pt"))
        again = M.load_best_model()
        for a, b in zip(net.state_dict().values(), again.state_dict().values()):
            assert torch.equal(a, b)
        torch.manual_seed(5)
        n7 = M.PolicyValueNet(5, 1, 8, name="model_7")
        M.save_model(n7, "model_7.h5")                             # extension is replaced by .pt
        assert M.model_name("LATEST") == "model_7"
        lat = M.load_latest_model()
        assert lat.name == "model_7" and torch.equal(lat.p_fc.weight, n7.p_fc.weight)
        assert M.load_model_by_name("model_7.h5").name == "model_7"
        # a Keras file that cannot be read must not degrade to random weights
        os.remove(os.path.join(conf['MODEL_DIR'], "best_model.pt"))
        open(os.path.join(conf['MODEL_DIR'], "best_model.h5"), "wb").write(b"\\x89HDF\\r\\n\\x1a\\n" + b"\\0" * 64)
        # (a) no HDF5 reader at all: refuse loudly; (b) a reader exists: the broken file is an error of the reader
        from sejonggo_amd import keras_import
        real = keras_import.h5_module

        def no_reader():
            raise ImportError("no HDF5 reader")
        keras_import.h5_module = no_reader
        try:
            with pytest.raises(RuntimeError, match="h5py"):
                M.load_best_model()
            with pytest.raises(RuntimeError, match="h5py"):
                M.model_name("BEST")
        finally:
            keras_import.h5_module = real
        try:
            real()
            with pytest.raises(Exception):
                M.load_best_model()
        except ImportError:
            pass
    finally:
        _restore(conf, old)


def test_latest_sym_compat_flag():
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.conf import conf
    old = conf.get('COMPAT_LATEST_SYM', True)
    try:
        conf['COMPAT_LATEST_SYM'] = True
        assert pq._kind("LATEST_SYM") == "BEST" and pq._kind("LATEST") == "LATEST"     # predicting_queue_worker.py:92
        assert pq._kind("LATEST_SYM", for_name=True) == "LATEST"
        conf['COMPAT_LATEST_SYM'] = False
        assert pq._kind("LATEST_SYM") == "LATEST"
        assert pq._kind("BEST_SYM") == pq._kind("BEST_NAME") == "BEST"
    finally:
        conf['COMPAT_LATEST_SYM'] = old


def test_simulation_workers_surface():
    from sejonggo_amd import simulation_workers as sw
    sw.init_simulation_workers_by_gpuid(3)
    assert sw.process_pool is not None
    got = []
    r = sw.process_pool.apply_async(lambda a, b: a + b, (1, 2), callback=got.append)
    assert r.get() == 3 and got == [3] and r.ready()
    assert sw.process_pool.map(lambda t: t * 2, [1, 2, 3]) == [2, 4, 6]
    q = sw.simulation_result_queue[5]
    assert q.empty()
    q.put(("node", [1]))
    assert q.get() == ("node", [1]) and q.empty()
    with pytest.raises(RuntimeError):
        q.get()
    sw.destroy_simulation_workers()
    assert sw.process_pool is None
    sw.destroy_simulation_workers()                     # idempotent, like the reference's `if process_pool is not None`
    sw.init_simulation_workers()
    assert sw.lock is not None and sw.process_pool is not None
    sw.destroy_simulation_workers()


def test_sample_dir_collision_bumps_game_number(tmp_path):
    """sgfsave.py:59-73: an existing move directory bumps the game number (for the rest of the game)."""
    from sejonggo_amd import sgfsave
    conf, old = _with_conf(tmp_path)
    try:
        S = 5
        gd = {'winner': 1, 'moves': [{'board': np.zeros((1, S, S, 17), np.int32), 'policy': np.zeros(S * S + 1), 'player': 1,
                                      'move_n': k, 'value': np.float32(0)} for k in range(2)]}
        os.makedirs(os.path.join(conf['SELF_PLAY_DIR'], "m", "game_00004", "move_000"))
        sgfsave.save_self_play_data("m", 4, gd)
        assert not os.path.exists(os.path.join(conf['SELF_PLAY_DIR'], "m", "game_00004", "move_000", "sample.h5"))
        assert os.path.isfile(os.path.join(conf['SELF_PLAY_DIR'], "m", "game_00005", "move_000", "sample.h5"))
        assert os.path.isfile(os.path.join(conf['SELF_PLAY_DIR'], "m", "game_00005", "move_001", "sample.h5"))
        sgfsave.save_game_data("m", 1, gd, game_name="eval_game")
        assert os.path.isfile(os.path.join(conf['GAMES_DIR'], "m", "eval_game_001", "move_001", "sample.h5"))
    finally:
        _restore(conf, old)


# ------------------------------------------------------------------------------------------------- GPU
SMALL = dict(SIZE=9, N_RESIDUAL_BLOCKS=2, N_GAMES=3, GAMES_PER_GPU=2, MCTS_SIMULATIONS=16, ENERGY=8, N_GAME_PROCESS=1,
             GPUs=[0], STOP_EXPLORATION=4, RESIGNATION_PERCENT=1.0)


@pytest.mark.gpu
def test_main_selfplay_end_to_end_by_fork(tmp_path):
    """sejonggo_amd.main_selfplay.main() from a fresh interpreter: the parent stays GPU-free, the worker is forked,
    loads model_1 onto the GPU itself, plays N_GAMES = 3 games and writes the reference's directory layout; the second
    loop iteration finds no new best model and stops (main_selfplay.py:20-23)."""
    out = _run(CONF_SNIPPET.format(root=ROOT, tmp=str(tmp_path)) + textwrap.dedent("""
        conf.update(%r)
        import multiprocessing
        from sejonggo_amd import _lib, main_selfplay
        main_selfplay.main()
        assert not _lib.gpu_runtime_initialised(), "the parent initialised the GPU"
        assert multiprocessing.get_start_method() == "fork"
        print("DONE")
    """ % (SMALL,)))
    assert "SELF-PLAYING BEST MODEL  model_1" in out and "No new best model for self-playing. Stopping.." in out
    assert "EXCEPTION" not in out, out
    for g in range(3):
        f = os.path.join(str(tmp_path), "selfplay", "model_1", "game_%05d" % g, "move_000", "sample.h5")
        assert os.path.isfile(f), (f, out)
    from tests.helpers import read_sample
    d = os.path.join(str(tmp_path), "selfplay", "model_1", "game_00001", "move_000")
    b, p, v = read_sample(os.path.join(d, "sample.h5"))
    assert b.shape == (1, 9, 9, 17) and b.dtype == np.float32 and p.shape == (82,) and v in (1.0, -1.0)
    assert b[0, :, :, :16].sum() == 0 and (b[0, :, :, 16] == 1).all()          # move 0: empty board, black to play
    assert abs(float(p.sum()) - 1.0) < 0.3 and (p >= 0).all()                  # priors (noise-mixed), not renormalised


@pytest.mark.gpu
def test_worker_started_by_spawn_when_parent_used_the_gpu(tmp_path):
    """A parent that has initialised the GPU (this pytest process) cannot fork GPU workers: start() switches to
    'spawn' and carries conf across."""
    import torch
    from sejonggo_amd import _lib
    from sejonggo_amd.selfplay_worker import NoModelSelfPlayWorker
    torch.zeros(1, device="cuda")
    assert _lib.gpu_runtime_initialised()
    conf, old = _with_conf(tmp_path, **dict(SMALL, N_GAMES=2))
    try:
        w = NoModelSelfPlayWorker(0)
        w.start()
        w.join(600)
        assert w.exitcode == 0
    finally:
        _restore(conf, old)
    for g in range(2):
        assert os.path.isfile(os.path.join(str(tmp_path), "selfplay", "model_1", "game_%05d" % g, "move_000", "sample.h5"))


@pytest.mark.gpu
def test_shipped_selfplay_runs_the_tower_kernel(tmp_path):
    """run_selfplay with the default loaders (no set_model_factory): the resident net must be the fused inference form
    whose 3x3 convolutions go through libsgo_hip.so (the benchmarked kernel is the shipped one)."""
    from sejonggo_amd import predicting_queue_worker as pq
    from sejonggo_amd.net import FusedInferenceNet
    from sejonggo_amd.selfplay_worker import run_selfplay
    conf, old = _with_conf(tmp_path, **dict(SMALL, N_GAMES=2))
    seen = []
    try:
        pq.set_model_factory(None)
        lib = _lib_counting()
        played = run_selfplay(0, "BEST_SYM", on_game=lambda g, gd: seen.append((g, len(gd['moves']))))
        net = pq.get_model("BEST_SYM", 0)
        assert isinstance(net, FusedInferenceNet) and net.name == "model_1" and net.fused_conv
        assert lib["calls"] > 0, "sgo_conv3x3_bias_act_dev was never called by the shipped path"
        assert played == 2 and sorted(g for g, _ in seen) == [0, 1] and all(n > 0 for _, n in seen)
    finally:
        _lib_counting(restore=True)
        pq.destroy_predicting_workers([0])
        _restore(conf, old)


_counting = {}


def _lib_counting(restore=False):
    """Counts calls of the fused convolution entry point made through net.FusedInferenceNet."""
    from sejonggo_amd import net as N
    if restore:
        if "orig" in _counting:
            N.FusedInferenceNet._conv = _counting.pop("orig")
        return None
    _counting["orig"] = N.FusedInferenceNet._conv
    _counting["calls"] = 0
    orig = _counting["orig"]

    def counted(self, x, w, b, pad, skip=None):
        if self.fused_conv:
            _counting["calls"] += 1
        return orig(self, x, w, b, pad, skip=skip)
    N.FusedInferenceNet._conv = counted
    return _counting


@pytest.mark.gpu
def test_selfplay_worker_sync_path_one_game_only(tmp_path):
    """SelfPlayWorker(gpuid, one_game_only=k) plays exactly game k with self_play.model_self_play
    (selfplay_worker.py:29-58, self_play.py:292-339) and stops when `forever` is off."""
    from sejonggo_amd.selfplay_worker import SelfPlayWorker
    conf, old = _with_conf(tmp_path, **dict(SMALL, N_GAMES=4, MCTS_SIMULATIONS=8, MCTS_BATCH_SIZE=4, N_RESIDUAL_BLOCKS=1))
    try:
        w = SelfPlayWorker(0, forever=False, one_game_only=2)
        w.start()
        w.join(900)
        assert w.exitcode == 0
    finally:
        _restore(conf, old)
    games = sorted(os.listdir(os.path.join(str(tmp_path), "selfplay", "model_1")))
    assert games == ["game_00002"]
    assert os.path.isfile(os.path.join(str(tmp_path), "selfplay", "model_1", "game_00002", "move_000", "sample.h5"))


def test_promote_best_model(tmp_path):
    """evaluator.py:50-85 bookkeeping: win files -> win rate -> the 55 % gate -> best model replaced, results cleaned."""
    import torch
    from sejonggo_amd import evaluator as ev, model as M
    conf, old = _with_conf(tmp_path, SIZE=5, N_RESIDUAL_BLOCKS=1, NET_CHANNELS=8, EVALUATE_MARGIN=.55)
    try:
        os.makedirs(conf['MODEL_DIR']); os.makedirs(conf['EVAL_DIR'])
        torch.manual_seed(1)
        M.save_model(M.PolicyValueNet(5, 1, 8, name="model_1"), "best_model")
        torch.manual_seed(2)
        cand = M.PolicyValueNet(5, 1, 8, name="model_2")
        M.save_model(cand, "model_2")
        for g in range(10):
            ev.save_eval_game("model_2", g, "model_2" if g < 5 else "model_1")       # 50 %: not enough
        assert ev.eval_statistic() == {"model_2": 0.5}
        assert ev.promote_best_model() is False and M.load_best_model().name == "model_1"
        ev.save_eval_game("model_2", 10, "model_2"); ev.save_eval_game("model_2", 11, "model_2")   # 7 / 12 = 58 %
        assert ev.promote_best_model() is True
        best = M.load_best_model()
        assert best.name == "model_2" and torch.equal(best.p_fc.weight, cand.p_fc.weight)
        assert os.listdir(conf['EVAL_DIR']) == []
    finally:
        _restore(conf, old)


def test_promoted_keras_file_is_not_shadowed_by_a_stale_checkpoint(tmp_path):
    """a fresh MODEL_DIR holds best_model.pt (create_initial_model); the reference's trainer then writes
    model_2.h5 and the evaluator promotes it.  load_best_model / model_name must resolve to the PROMOTED model, and a
    best_model.h5 dropped next to best_model.pt by the reference's own evaluator (a plain copy) wins by being newer."""
    import time
    import torch
    from sejonggo_amd import evaluator as ev, keras_import as ki, model as M
    try:
        ki.h5_module()
    except ImportError:
        pytest.skip("no HDF5 reader (h5py / libhdf5) on this box")
    conf, old = _with_conf(tmp_path, SIZE=5, N_RESIDUAL_BLOCKS=1, NET_CHANNELS=8, EVALUATE_MARGIN=.55)
    try:
        os.makedirs(conf['MODEL_DIR']); os.makedirs(conf['EVAL_DIR'])
        assert M.load_best_model().name == "model_1"                               # writes model_1.pt + best_model.pt
        assert os.path.isfile(os.path.join(conf['MODEL_DIR'], "best_model.pt"))
        torch.manual_seed(5)
        cand = M.PolicyValueNet(5, 1, 8, name="model_2").eval()
        ki.save_keras_h5(os.path.join(conf['MODEL_DIR'], "model_2.h5"), cand)      # a .h5-only candidate
        for g in range(10):
            ev.save_eval_game("model_2", g, "model_2")
        assert ev.promote_best_model() is True
        assert sorted(f for f in os.listdir(conf['MODEL_DIR']) if f.startswith("best_model")) == ["best_model.h5"]
        assert M.model_name("BEST") == "model_2"
        best = M.load_best_model()
        assert best.name == "model_2" and torch.allclose(best.p_fc.weight, cand.p_fc.weight)
        # both extensions present (the reference's evaluator copied a .h5 beside our .pt): the newer file wins
        M.save_model(M.PolicyValueNet(5, 1, 8, name="model_1"), "best_model")       # -> best_model.pt, drops the .h5
        assert M.model_name("BEST") == "model_1"
        time.sleep(0.02)
        ki.save_keras_h5(os.path.join(conf['MODEL_DIR'], "best_model.h5"), cand)
        assert M.model_name("BEST") == "model_2" and M.load_best_model().name == "model_2"
    finally:
        _restore(conf, old)


@pytest.mark.gpu
def test_evaluation_worker_on_the_device(tmp_path):
    """evaluate_worker.NoModelEvaluateWorker's body: best vs latest as concurrent two-model slots of the device engine, eval
    games saved as training data + one result file per game, then the 55 % gate (evaluator.py:50-80)."""
    import torch
    from sejonggo_amd import evaluator as ev, model as M, predicting_queue_worker as pq
    from sejonggo_amd.evaluate_worker import run_evaluation
    conf, old = _with_conf(tmp_path, SIZE=9, N_RESIDUAL_BLOCKS=1, NET_CHANNELS=32, MCTS_SIMULATIONS=16, ENERGY=8, GPUs=[0],
                           EVALUATE_N_GAMES=6, GAMES_PER_GPU=4, COMPAT_LATEST_SYM=False, EVALUATE_MARGIN=.55)
    try:
        os.makedirs(conf['MODEL_DIR']); os.makedirs(conf['EVAL_DIR'])
        torch.manual_seed(1)
        M.save_model(M.PolicyValueNet(9, 1, 32, name="model_1"), "best_model")
        M.save_model(M.PolicyValueNet(9, 1, 32, name="model_1"), "model_1")
        torch.manual_seed(2)
        M.save_model(M.PolicyValueNet(9, 1, 32, name="model_2"), "model_2")
        pq.set_model_factory(None)
        wins, total = run_evaluation(0, engine_kwargs={"num_moves": 30})
        assert total == 6 and 0 <= wins <= 6
        stat = ev.eval_statistic()
        assert set(stat) == {"model_2"} and abs(stat["model_2"] - wins / 6.0) < 1e-9
        for g in range(6):
            d = os.path.join(conf['EVAL_DIR'], "model_2", "game_%03d" % g)
            names = os.listdir(d)
            assert len(names) == 1 and names[0] in ("model_1", "model_2", "None")
            assert os.path.isfile(os.path.join(conf['GAMES_DIR'], "model_2", "eval_game_%03d" % g, "move_000", "sample.h5"))
        promoted = ev.promote_best_model()
        assert promoted == (wins / 6.0 > .55)
        assert M.load_best_model().name == ("model_2" if promoted else "model_1")
        # best == latest: the worker quits like the reference (evaluate_worker.py:153-154)
        os.remove(os.path.join(conf['MODEL_DIR'], "model_2.pt"))
        pq.destroy_predicting_workers([0])
        if not promoted:
            assert run_evaluation(0) == (0, 0)
    finally:
        pq.destroy_predicting_workers([0])
        _restore(conf, old)


@pytest.mark.gpu
def test_driver_smoke_entry():
    """__graft_entry__.smoke() -- what the driver runs before the bench -- stays runnable."""
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_distributed_selfplay_gathers_tuples_to_rank0(tmp_path, world, backend):
    """sejonggo_amd.dist_selfplay: N ranks (here 2 over gloo sharing the one GPU of the box, and 1 over RCCL), games sharded
    g -> rank g mod N, weights broadcast from rank 0, finished games gathered to rank 0 as packed (s, pi, z) tuples and
    written there in the reference's sample layout.  Every game number must come out exactly once and complete."""
    import json
    from tests.helpers import read_sample
    over = dict(MODEL_DIR=str(tmp_path / "models"), SELF_PLAY_DIR=str(tmp_path / "selfplay"), EVAL_DIR=str(tmp_path / "eval"),
                LOG_DIR=str(tmp_path / "logs"), TMP_DIR=str(tmp_path / "tmp"), GAMES_DIR=str(tmp_path / "eval"),
                SIZE=9, N_RESIDUAL_BLOCKS=1, NET_CHANNELS=32, N_GAMES=7, GAMES_PER_GPU=3, MCTS_SIMULATIONS=16, ENERGY=8,
                STOP_EXPLORATION=4, RESIGNATION_PERCENT=1.0, NUM_MOVES=12)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", SGO_CONF_JSON=json.dumps(over), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "sejonggo_amd.dist_selfplay", "--gpus", str(world), "--backend", backend, "--sync-every", "3"],
                       capture_output=True, text=True, timeout=900, env=env, cwd="/tmp")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    out = r.stdout
    import re
    # by pattern, not by line: the ranks share one pipe, and the collective library's own start-up messages do not always end
    # their line before a rank's summary lands on it
    played = sorted(int(n) for _, n in re.findall(r"rank (\d+): (\d+) games played", out))
    assert sum(played) == 7 and len(played) == world, out[-2000:]
    if world == 2:
        assert played == [3, 4]                                  # games 1,3,5 on rank 1; 0,2,4,6 on rank 0
    assert "%d positions written" % (7 * 12) in out
    root = os.path.join(str(tmp_path), "selfplay", "model_1")
    assert sorted(os.listdir(root)) == ["game_%05d" % g for g in range(7)]
    for g in range(7):
        moves = sorted(os.listdir(os.path.join(root, "game_%05d" % g)))
        assert moves == ["move_%03d" % k for k in range(12)]
        b, p, v = read_sample(os.path.join(root, "game_%05d" % g, "move_000", "sample.h5"))
        assert b.shape == (1, 9, 9, 17) and b[0, :, :, :16].sum() == 0 and (b[0, :, :, 16] == 1).all()
        assert p.shape == (82,) and v in (1.0, -1.0)
        b5, _, _ = read_sample(os.path.join(root, "game_%05d" % g, "move_005", "sample.h5"))
        assert (b5[0, :, :, 16] == -1).all() and 1 <= b5[0, :, :, :2].sum() <= 5      # white to play after five plies


@pytest.mark.gpu
@pytest.mark.parametrize("tower", [-1, 2], ids=["default_tower", "packed_tower"])
def test_bench_stdout_is_one_json_line(tower):
    """The driver reads ONE JSON line from bench.py's stdout: whatever libraries print from C (RCCL's banner) must not land
    there.  Small configuration (9x9, the real 256-channel tower with enough launches for the 1-in-16 sampled
    launch timing, so the roofline object names the tower kernel)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--size", "9", "--games", "64", "--sims", "64", "--blocks", "4",
                        "--steps", "1", "--warmup", "1", "--cpu-baseline", "0", "--saturated", "0", "--steady-state", "0",
                        "--tower-kernel", str(tower)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1, lines[:5]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["value"] > 0
    # bench.py --tower-kernel 2 runs the step on k_conv4r (net.use_packed_tower) and says so in the roofline object
    assert ("k_conv4r" if tower == 2 else "k_conv4w") in out["roofline"]["kernel"] and 0 < out["roofline"]["frac"] < 1
